"""Synthetic checkpoints in the form the published ones have on disk (PyTorch / torchtune key names and conv layouts), i.e. what ``Model.sanitize`` of
each family receives.  Shared by tests/golden/make_reference_fixtures.py (feeds them to the reference's own ``sanitize``) and the tests that feed them to
this package's ``sanitize`` (tests/test_reference_fixtures_cpu.py; the loader tests build the same forms inline)."""
import re

import torch


def kokoro_checkpoint(w):
    """mlx_audio_amd.tts.models.kokoro.synthetic weights -> the HF checkpoint's form: PyTorch conv layout (out, in, K) for the keys sanitize transposes,
    torch LSTM names, LayerNorm gamma / beta in the text encoder, plus a ``position_ids`` buffer that sanitize drops."""
    names = {"Wx_forward": "weight_ih_l0", "Wh_forward": "weight_hh_l0", "bias_ih_forward": "bias_ih_l0", "bias_hh_forward": "bias_hh_l0",
             "Wx_backward": "weight_ih_l0_reverse", "Wh_backward": "weight_hh_l0_reverse", "bias_ih_backward": "bias_ih_l0_reverse",
             "bias_hh_backward": "bias_hh_l0_reverse"}
    out = {}
    for k, v in w.items():
        base, _, suf = k.rpartition(".")
        if suf in names and k.startswith(("text_encoder", "predictor")):
            out[f"{base}.{names[suf]}"] = v
        elif k.startswith("text_encoder.cnn") and re.search(r"\.1\.(weight|bias)$", k):
            out[base + (".gamma" if suf == "weight" else ".beta")] = v
        elif k.endswith(("F0_proj.weight", "N_proj.weight")) or ("noise_convs" in k and k.endswith(".weight")) or (k.endswith("weight_v") and v.dim() == 3):
            out[k] = v.permute(0, 2, 1).contiguous()
        else:
            out[k] = v
    out["bert.embeddings.position_ids"] = torch.arange(8)[None]
    return out


def csm_checkpoint(w):
    """mlx_audio_amd.tts.models.sesame.engine.make_csm_weights (canonical stack names) -> torchtune names as sesame/csm-1b stores them."""
    tt = {"wq": "attn.q_proj", "wk": "attn.k_proj", "wv": "attn.v_proj", "wo": "attn.output_proj", "w_gate": "mlp.w1", "w_down": "mlp.w2", "w_up": "mlp.w3"}
    ck = {}
    for k, v in w.items():
        m = re.match(r"^(backbone|decoder)\.layers\.(\d+)\.(\w+)\.weight$", k)
        if m and m.group(3) in tt:
            ck[f"{m.group(1)}.layers.{m.group(2)}.{tt[m.group(3)]}.weight"] = v
        elif m and m.group(3) in ("attn_norm", "mlp_norm"):
            ck[f"{m.group(1)}.layers.{m.group(2)}.{'sa_norm' if m.group(3) == 'attn_norm' else 'mlp_norm'}.scale"] = v
        elif k.endswith("final_norm.weight"):
            ck[k.replace("final_norm.weight", "norm.scale")] = v
        else:
            ck[k] = v
    return ck


def qwen3_codec_checkpoint(cw):
    """mlx_audio_amd.tts.models.qwen3_tts.synthetic.make_codec_decoder_weights -> the speech_tokenizer/model.safetensors form: ``decoder.`` prefix, codebooks as
    embedding_sum / cluster_usage, PyTorch Conv1d (out, in, K) and ConvTranspose1d (in, out, K) layouts, plus an encoder key that sanitize has to place."""
    ck = {}
    for k, v in cw.items():
        if k.endswith("codebook.embed.weight"):
            base = k[: -len(".codebook.embed.weight")]
            ck[f"decoder.{base}._codebook.cluster_usage"] = torch.full((v.shape[0],), 2.0)
            ck[f"decoder.{base}._codebook.embedding_sum"] = 2.0 * v
        elif v.dim() == 3 and (("upsample" in k and ".0.conv.weight" in k) or re.search(r"decoder\.\d+\.block\.1\.conv\.weight", k)):
            ck["decoder." + k] = v.permute(2, 0, 1).contiguous()
        elif v.dim() == 3:
            ck["decoder." + k] = v.permute(0, 2, 1).contiguous()
        else:
            ck["decoder." + k] = v
    return ck


def summary(d):
    """key -> [shape, sum, sum of squares] (float64): enough to tell two sanitized dicts apart, small enough to commit."""
    return {k: [list(v.shape), float(torch.as_tensor(v).double().sum()), float((torch.as_tensor(v).double() ** 2).sum())] for k, v in sorted(d.items())}


def fake_kitten_wave(n_tokens: int, speed: float):
    """A deterministic stand-in for a synthesised chunk (float32 numpy, 24 kHz): speech-like noise, then 50 ms of near-silence and a short spurt at the
    end -- the pattern the tail trimmer of KittenTTS's ``generate`` looks for -- with length and seed depending on the inputs."""
    import numpy as np

    rng = np.random.default_rng(1000 + n_tokens)
    n = int(24000 * (0.4 + 0.01 * n_tokens) / max(speed, 0.25))
    env = np.minimum(1.0, np.arange(n) / 2400.0) * (0.6 + 0.4 * np.sin(np.arange(n) / 3000.0) ** 2)
    body = (0.2 * rng.standard_normal(n) * env).astype(np.float32)
    gap = (1e-4 * rng.standard_normal(1200)).astype(np.float32)
    spurt = (0.15 * rng.standard_normal(700)).astype(np.float32)
    return np.concatenate([body, gap, spurt]).astype(np.float32)


KITTEN_GENERATE_CASES = [
    dict(text="Hello there, this is a test", kw=dict()),
    dict(text="The quick brown fox jumps. Over the lazy dog! And then it sleeps for a while", kw=dict(chunk_size=30)),
    dict(text="One two three. Four five six? Seven eight", kw=dict(chunk_size=16, crossfade_ms=0, fade_out_ms=0, tail_silence_ms=0)),
    dict(text="Short one. Another short one", kw=dict(chunk_size=12, crossfade_ms=50, fade_out_ms=600, tail_silence_ms=100, speed=1.3)),
]


class WhisperCodec:
    """Deterministic text <-> token stand-in shared by the reference-side tokenizer stub and this package's ``get_tokenizer(codec=...)``."""

    def encode(self, text):
        return [100 + (ord(c) % 50) for c in text]

    def decode(self, toks):
        return "".join(chr(ord("a") + (int(t) % 26)) for t in toks)


TB = 50364  # timestamp_begin of the multilingual vocabulary

# Scripts for ``Model.generate`` with ``decode`` stubbed: every entry of "script" is the DecodingResult of one decode call, either a dict or a
# {temperature: dict} table (temperature fallback).  "frames" = content frames of the (ramp) mel, "kw" = generate() options.
WHISPER_GENERATE_CASES = [
    dict(name="two_windows", frames=4000, kw=dict(temperature=0.0),
         script=[dict(tokens=[TB, 200, 201, TB + 1500]), dict(tokens=[TB, 300, TB + 100])]),
    dict(name="fallback", frames=3000, kw=dict(),
         script=[{0.0: dict(tokens=[TB, 200, TB + 1500], compression_ratio=3.0), 0.2: dict(tokens=[TB, 210, TB + 1500], avg_logprob=-1.5),
                  0.4: dict(tokens=[TB, 220, 221, TB + 1500], avg_logprob=-0.3)}] * 1),
    dict(name="no_speech_skip", frames=6000, kw=dict(temperature=0.0),
         script=[dict(tokens=[TB, 200, TB + 1500], no_speech_prob=0.9, avg_logprob=-2.0), dict(tokens=[TB, 300, 301, TB + 1500])]),
    dict(name="consecutive_timestamps", frames=3000, kw=dict(temperature=0.0),
         script=[dict(tokens=[TB, 100, 101, TB + 200, TB + 200, 102, TB + 400, TB + 400, 103, 104, TB + 900]),
                 dict(tokens=[TB, 105, TB + 600])]),
    dict(name="consecutive_ending_pair", frames=3000, kw=dict(temperature=0.0),
         script=[dict(tokens=[TB, 100, TB + 200, TB + 200, 102, TB + 400, TB + 400]), dict(tokens=[TB, 105, TB + 1100])]),
    dict(name="prompt_and_no_conditioning", frames=5000, kw=dict(temperature=0.0, condition_on_previous_text=False, initial_prompt="hello there"),
         script=[dict(tokens=[TB, 200, TB + 1500]), dict(tokens=[TB, 300, TB + 1000])]),
    dict(name="conditioning_resets_at_high_temperature", frames=7000, kw=dict(temperature=(0.0, 0.8)),
         script=[{0.0: dict(tokens=[TB, 200, TB + 1500], avg_logprob=-3.0), 0.8: dict(tokens=[TB, 201, TB + 1500])}, dict(tokens=[TB, 300, TB + 1500]),
                 dict(tokens=[TB, 400, TB + 500])]),
    dict(name="clip_timestamps", frames=4000, kw=dict(temperature=0.0, clip_timestamps="5,12,20,26"),
         script=[dict(tokens=[TB, 200, TB + 350]), dict(tokens=[TB, 300, TB + 300])]),
    dict(name="no_timestamps", frames=3500, kw=dict(temperature=0.0, return_timestamps=False),
         script=[dict(tokens=[200, 201, 202]), dict(tokens=[300])]),
]
