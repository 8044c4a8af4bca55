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


def qwen3_tokenizer_encoder_checkpoint(mw, n_layers, nq):
    """Mimi-named ENCODER weights (mlx_audio_amd.codec.models.mimi.make_mimi_encoder_weights + the codebooks of make_mimi_decoder_weights) -> the
    HuggingFace form of the Qwen3-TTS speech tokenizer's encoder half (``encoder.`` prefix: SeanetEncoder layers by index with ``block.1`` / ``block.3``,
    separate q / k / v projections, ``embed_sum`` codebooks, PyTorch Conv1d layouts, plus an ``initialized`` buffer and an unused layer that sanitize drops)."""
    ck = {}
    pt = lambda v: v.permute(0, 2, 1).contiguous() if v.dim() == 3 else v
    seanet = {"encoder.init_conv1d.conv.conv": "encoder.encoder.layers.0.conv", "encoder.final_conv1d.conv.conv": "encoder.encoder.layers.14.conv"}
    for i in range(4):
        seanet[f"encoder.layers.{i}.residuals.0.block.0.conv.conv"] = f"encoder.encoder.layers.{1 + 3 * i}.block.1.conv"
        seanet[f"encoder.layers.{i}.residuals.0.block.1.conv.conv"] = f"encoder.encoder.layers.{1 + 3 * i}.block.3.conv"
        seanet[f"encoder.layers.{i}.downsample.conv.conv"] = f"encoder.encoder.layers.{3 + 3 * i}.conv"
    for src, dst in seanet.items():
        for suf in ("weight", "bias"):
            if f"{src}.{suf}" in mw:
                ck[f"{dst}.{suf}"] = pt(mw[f"{src}.{suf}"])
    tr = {"self_attn.out_proj.weight": "self_attn.o_proj.weight", "gating.linear1.weight": "mlp.fc1.weight", "gating.linear2.weight": "mlp.fc2.weight",
          "norm1.weight": "input_layernorm.weight", "norm1.bias": "input_layernorm.bias", "norm2.weight": "post_attention_layernorm.weight",
          "norm2.bias": "post_attention_layernorm.bias", "layer_scale_1.scale": "self_attn_layer_scale.scale", "layer_scale_2.scale": "mlp_layer_scale.scale"}
    for i in range(n_layers):
        p = f"encoder_transformer.transformer.layers.{i}."
        ip = mw[p + "self_attn.in_proj.weight"]
        d = ip.shape[0] // 3
        for j, nm in enumerate(("q", "k", "v")):
            ck[f"encoder.encoder_transformer.layers.{i}.self_attn.{nm}_proj.weight"] = ip[j * d:(j + 1) * d].contiguous()
        for src, dst in tr.items():
            ck[f"encoder.encoder_transformer.layers.{i}.{dst}"] = mw[p + src]
    ck["encoder.downsample.conv.weight"] = pt(mw["downsample.conv.conv.conv.weight"])
    for half, hf, n in (("rvq_first", "semantic_residual_vector_quantizer", 1), ("rvq_rest", "acoustic_residual_vector_quantizer", nq - 1)):
        ck[f"encoder.quantizer.{hf}.input_proj.weight"] = pt(mw[f"quantizer.{half}.input_proj.weight"])
        ck[f"encoder.quantizer.{hf}.output_proj.weight"] = pt(mw[f"quantizer.{half}.output_proj.weight"])
        for i in range(n):
            ck[f"encoder.quantizer.{hf}.layers.{i}.codebook.embed_sum"] = mw[f"quantizer.{half}.vq.layers.{i}.codebook.embedding_sum"]
            ck[f"encoder.quantizer.{hf}.layers.{i}.codebook.cluster_usage"] = mw[f"quantizer.{half}.vq.layers.{i}.codebook.cluster_usage"]
            ck[f"encoder.quantizer.{hf}.layers.{i}.codebook.initialized"] = torch.ones(1)
    ck["encoder.encoder.layers.2.conv.weight"] = torch.zeros(3, 3, 3)   # an index the map does not name (ELU in the HF module list): dropped
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


# ---- Qwen3-TTS prompt assembly (make_reference_fixtures.run_qwen3_inputs and tests/test_tts_prompt_assembly_cpu.py share these)
QWEN3_TEXT_VOCAB, QWEN3_CODEC_VOCAB = 400, 120


class QwenCharTokenizer:
    """One id per character: the assembly only slices the id list ([:3] role, [3:4] first text token, [4:-5] trailing), so any tokenizer exercises it."""

    def __init__(self, vocab=None):
        self.vocab = vocab or QWEN3_TEXT_VOCAB

    def encode(self, text):
        return [7 + (ord(c) * 31) % (self.vocab - 10) for c in text]


def qwen3_input_config():
    from types import SimpleNamespace

    talker = SimpleNamespace(spk_id={"vivian": 101, "dylan": 102, "eric": 103}, spk_is_dialect={"vivian": False, "dylan": "beijing_dialect", "eric": "martian"},
                             codec_language_id={"english": 50, "chinese": 55, "beijing_dialect": 74}, codec_nothink_id=40, codec_think_id=41,
                             codec_think_bos_id=42, codec_think_eos_id=43, codec_pad_id=44, codec_bos_id=45, codec_eos_token_id=46)
    return SimpleNamespace(talker_config=talker, tts_bos_token_id=391, tts_eos_token_id=392, tts_pad_token_id=393)


QWEN3_INPUT_CASES = [
    dict(text="Hello world.", language="auto", speaker=None, instruct=None),
    dict(text="Hello.", language="English", speaker="Vivian", instruct=None),
    dict(text="Ni hao ma", language="auto", speaker="dylan", instruct=None),                       # dialect override of the language id
    dict(text="Ni hao", language="chinese", speaker="Dylan", instruct="Speak slowly and warmly"),
    dict(text="Hi there", language="klingon", speaker=None, instruct=None),                       # unknown language: no-think prefix
    dict(text="Hi", language="english", speaker="nobody", instruct="Whisper"),                    # unknown speaker: no speaker slot
    dict(text="Guten Tag", language="english", speaker="eric", instruct=None),                    # dialect name without a language id: ignored
    dict(text="Bonjour", language="auto", speaker="vivian", instruct=None),                       # spk_is_dialect False
]
QWEN3_BATCH_CASE = dict(texts=["Hello world, this is the long one.", "Hi", "Ni hao ma"], language="auto", speakers=["vivian", None, "dylan"],
                        instructs=[None, "Be brief", None])


# ---- Qwen3-TTS voice cloning: in-context prompts, x-vector prompts, decode behind the reference codes, the batch's shared-reference rules
# (make_reference_fixtures.run_qwen3_icl and tests/test_tts_prompt_assembly_cpu.py share these stand-ins and case tables)
QWEN3_ICL_GROUPS, QWEN3_ICL_CP_VOCAB, QWEN3_ICL_UP = 4, 64, 8


def qwen3_fake_clip(n, seed):
    import numpy as _np

    return (_np.random.default_rng(seed).standard_normal(n) * 0.1).astype(_np.float32)


def qwen3_fake_codes(audio):
    """Stand-in for ``speech_tokenizer.encode``: [1, 1, n] (any leading ones) -> codes [1, groups, n // 50], values < 40, some first codes 0."""
    import numpy as _np

    a = _np.asarray(audio, dtype=_np.float64).reshape(-1)
    T = len(a) // 50
    base = _np.floor(_np.abs(a[:T * 50].reshape(T, 50)).sum(1) * 40).astype(_np.int64)
    return _np.stack([(base + 11 * g) % 40 for g in range(QWEN3_ICL_GROUPS)], 0)[None].astype(_np.int64)


def qwen3_fake_xvector(audio, H):
    import numpy as _np

    a = _np.asarray(audio, dtype=_np.float64).reshape(-1)
    return _np.cos(_np.arange(1, H + 1) * (1.0 + float(_np.abs(a).sum())))[None].astype(_np.float32)


def qwen3_fake_decode(codes):
    """Stand-in for ``speech_tokenizer.decode``: codes [1, T, groups] -> (audio [1, T * up] that depends on every code, valid sample counts [1])."""
    import numpy as _np

    c = _np.asarray(codes, dtype=_np.int64)
    per_frame = (c * _np.arange(1, c.shape[2] + 1)).sum(2).astype(_np.float32) / 100.0          # [1, T]
    audio = (per_frame[:, :, None] + _np.arange(QWEN3_ICL_UP, dtype=_np.float32) / 1000.0).reshape(1, -1)
    return audio.astype(_np.float32), ((c[..., 0] > 0).sum(1) * QWEN3_ICL_UP).astype(_np.int64)


def qwen3_icl_config(kind="base"):
    c = qwen3_input_config()
    c.talker_config.num_code_groups = QWEN3_ICL_GROUPS
    c.tts_model_type = kind
    return c


# clip: (samples, seed, leading axes added before the call); xvec: whether the model has a speaker encoder
QWEN3_ICL_CASES = [
    dict(text="Hello there.", ref_text="Reference words", language="auto", clip=(400, 1, 0), xvec=True),
    dict(text="Bonjour", ref_text="Reference words", language="English", clip=(400, 1, 0), xvec=True),      # same clip + transcript: cache hit; language id
    dict(text="A", ref_text="Other transcript", language="klingon", clip=(650, 2, 1), xvec=True),            # [1, n] clip; unknown language
    dict(text="No x-vector here", ref_text="Ref", language="chinese", clip=(300, 3, 0), xvec=False),
]
QWEN3_ICL_BATCH = dict(texts=["First sentence of the batch.", "Two", "And a third one"], ref_text="Shared reference", language="auto", clip=(500, 4, 0))
# plain prompts with a clip but no transcript: the x-vector takes the speaker slot (also over a named speaker); ignored without a speaker encoder
QWEN3_XVEC_CASES = [
    dict(text="Hello world.", language="auto", speaker=None, clip=(400, 1, 0), xvec=True),
    dict(text="Hello.", language="English", speaker="Vivian", clip=(350, 5, 0), xvec=True),
    dict(text="Hello.", language="English", speaker="Vivian", clip=(350, 5, 0), xvec=False),
]
# (generated frames, reference frames): decode behind the reference codes, trim to the valid length, cut the reference's share
QWEN3_ICL_DECODE_CASES = [(5, 3), (1, 9), (7, 1), (0, 4)]


def qwen3_icl_decode_case(n_gen, n_ref):
    import numpy as _np

    g = _np.random.default_rng(100 * n_gen + n_ref)
    gen = g.integers(0, 3, size=(n_gen, QWEN3_ICL_GROUPS)).astype(_np.int64)        # zeros among the first codes: "invalid" frames
    ref = g.integers(1, 40, size=(1, QWEN3_ICL_GROUPS, n_ref)).astype(_np.int64)
    return gen, ref


_A, _B = "clip-a.wav", "clip-b.wav"
QWEN3_SHARED_REF_CASES = [
    dict(n=2),
    dict(n=2, ref_audio=_A, ref_text="t"),
    dict(n=2, ref_audio=_A),
    dict(n=2, ref_text="t"),
    dict(n=2, ref_audios=[_A, _A], ref_texts=["t", "t"]),
    dict(n=2, ref_audios=[_A, _B], ref_texts=["t", "t"]),
    dict(n=2, ref_audios=[_A, _A], ref_texts=["t", "u"]),
    dict(n=2, ref_audios=[_A, None], ref_texts=["t", "t"]),
    dict(n=2, ref_audios=[_A], ref_texts=["t", "t"]),
    dict(n=2, ref_audios=[None, None], ref_texts=[None, None]),
    dict(n=2, ref_audio=_B, ref_audios=[_A, _A], ref_texts=["t", "t"]),
    dict(n=2, ref_audio=_A, ref_text="u", ref_audios=[_A, _A], ref_texts=["t", "t"]),
    dict(n=3, ref_audio=_A, ref_texts=["t", "t", "t"]),
]


def qwen3_shared_ref_outcome(fn, case):
    kw = {k: v for k, v in case.items() if k != "n"}
    try:
        a, t = fn(case["n"], **kw)
        return ["ok", a, t]     # the caller patches load_audio to return "loaded:<path>"
    except ValueError as e:
        return ["error", str(e)]


QWEN3_SUPPORTS_BATCH_CASES = [
    dict(kind="base", enc=True, kw=dict(ref_audio="x", ref_text="t")),
    dict(kind="base", enc=False, kw=dict(ref_audio="x", ref_text="t")),
    dict(kind="base", enc=True, kw=dict(ref_audio="x")),
    dict(kind="base", enc=True, kw=dict(ref_text="t")),
    dict(kind="base", enc=True, kw=dict(ref_audio="x", ref_text="t", voice="vivian")),
    dict(kind="base", enc=True, kw=dict(ref_audio="x", ref_text="t", instruct="slow")),
    dict(kind="base", enc=True, kw=dict(ref_audio="x", ref_text="t", stream=True)),
    dict(kind="base", enc=True, kw=dict(ref_audio="x", ref_text="t", speed=1.2)),
    dict(kind="custom_voice", enc=True, kw=dict(ref_audio="x", ref_text="t", voice="vivian")),
    dict(kind="custom_voice", enc=True, kw=dict(voice="vivian")),
    dict(kind="voice_design", enc=True, kw=dict(instruct="deep")),
    dict(kind="base", enc=True, kw=dict()),
    dict(kind="base", enc=None, kw=dict(ref_audio="x", ref_text="t")),     # no speech tokenizer at all
]


# ---- CSM prompt frames + generate bookkeeping (make_reference_fixtures.run_csm_generate and tests/test_tts_prompt_assembly_cpu.py)
CSM_CODEBOOKS = 4


class CsmCharTokenizer:
    def ids(self, text):
        return [3 + (ord(c) * 17) % 200 for c in text]


def csm_fake_codes(audio_1d):
    """Stand-in for ``Mimi.encode``: (K, T) codes that depend on the samples, T = len // 5."""
    import numpy as _np

    a = _np.asarray(audio_1d, dtype=_np.float64).reshape(-1)
    T = len(a) // 5
    base = _np.floor(_np.abs(a[:T * 5].reshape(T, 5)).sum(1) * 100).astype(_np.int64)
    return _np.stack([(base + 7 * k) % 50 + 1 for k in range(CSM_CODEBOOKS)], 0).astype(_np.int32)


def csm_audio(n, seed):
    import numpy as _np

    return _np.random.default_rng(seed).standard_normal(n).astype(_np.float32)


# "frames": how many non-EOS frames the scripted model emits per prompt (then an all-zero frame unless the frame budget ends the loop first)
CSM_GENERATE_CASES = [
    dict(name="plain", text="Hello there.", kw=dict(speaker=0), cfg=dict(speaker_prefix_space=False, voice_match=True), frames=[5]),
    dict(name="prefix_space_two_prompts", text="  First line.\n\nSecond line.", kw=dict(speaker=3), cfg=dict(speaker_prefix_space=True, voice_match=True),
         frames=[4, 2]),
    dict(name="context_voice_match", text="And then more.", kw=dict(speaker=1, context=[(1, "I said this", (23, 1)), (0, "ignored second", (11, 2))]),
         cfg=dict(speaker_prefix_space=False, voice_match=True), frames=[3]),
    dict(name="context_no_voice_match", text="A reply.", kw=dict(speaker=1, context=[(1, "I said this", (23, 1)), (0, "You said that", (11, 2))], voice_match=False),
         cfg=dict(speaker_prefix_space=False, voice_match=True), frames=[3]),
    dict(name="ref_audio", text="Clone me.", kw=dict(speaker=2, ref_audio=(31, 5), ref_text="Reference words"), cfg=dict(speaker_prefix_space=True, voice_match=False),
         frames=[2]),
    dict(name="stream", text="Streaming output.", kw=dict(speaker=0, stream=True, streaming_interval=0.5), cfg=dict(speaker_prefix_space=False, voice_match=True),
         frames=[15]),
    dict(name="budget", text="Cut short.", kw=dict(speaker=0, max_audio_length_ms=400), cfg=dict(speaker_prefix_space=False, voice_match=True), frames=[50]),
    dict(name="list_of_prompts_no_split", text=["One\nstill one", "Two"], kw=dict(speaker=0), cfg=dict(speaker_prefix_space=False, voice_match=True), frames=[1, 0]),
]


def csm_frame(i, j):
    """Scripted frame j (non-zero) of prompt i."""
    return [1 + (5 * i + 3 * j + k) % 40 for k in range(CSM_CODEBOOKS)]

# special codec ids inside the tiny talker's 1200-entry vocabulary (run_qwen3_generate_loop and its test)
QWEN3_LOOP_CODEC_IDS = dict(codec_think_id=1154, codec_nothink_id=1155, codec_think_bos_id=1156, codec_think_eos_id=1157, codec_pad_id=1148, codec_bos_id=1149)


# ---- Kokoro pipeline chunking (make_reference_fixtures.run_kokoro_pipeline and tests/test_kokoro_pipeline_cpu.py)
class FakeMToken:
    """What the pipeline reads from a ``misaki.en.MToken``."""

    def __init__(self, text, phonemes, whitespace):
        self.text, self.phonemes, self.whitespace = text, phonemes, whitespace
        self.start_ts = self.end_ts = None


def kokoro_token_stream(kind, n_words):
    """A deterministic English token list: words with 1..9 phonemes (some with a flap, some unpronounceable), punctuation per ``kind``."""
    toks = []
    for i in range(n_words):
        ph = "".join("aɾbdefgik"[(i * 7 + j * 3) % 9] for j in range(1 + (i * 5) % 9))
        if i % 23 == 11:
            ph = None                                   # out-of-dictionary word without fallback
        mark = None
        if kind == "sentences" and i % 17 == 16:
            mark = ".!?…"[(i // 17) % 4]
        elif kind == "clauses" and i % 13 == 12:
            mark = ":;"[(i // 13) % 2]
        elif kind == "commas" and i % 9 == 8:
            mark = ",—"[(i // 9) % 2]
        elif kind == "mixed":
            if i % 41 == 40:
                mark = "."
            elif i % 11 == 10:
                mark = ","
        elif kind == "quoted" and i % 19 == 18:
            mark = "!"
        toks.append(FakeMToken(f"w{i}", ph, "" if mark else " "))
        if mark:
            toks.append(FakeMToken(mark, mark, "" if kind == "quoted" else " "))
            if kind == "quoted":
                toks.append(FakeMToken("”", "”", " "))
    return toks


def kokoro_fake_durations(ps):
    return [5] + [1 + (ord(c) * 7) % 4 for c in ps] + [3]


def kokoro_spanish_g2p(text):
    """Stand-in for an espeak G2P: one phoneme per letter, doubled vowels (so that long sentences exceed 510 phonemes); some calls return a tuple."""
    ps = "".join(c + c if c in "aeiou" else c for c in text.lower())
    return (ps, None) if len(text) % 2 else ps


KOKORO_PIPELINE_CASES = [
    dict(name="en_sentences", lang="a", tokens=("sentences", 420)),
    dict(name="en_clauses", lang="a", tokens=("clauses", 330)),
    dict(name="en_commas", lang="b", tokens=("commas", 300)),
    dict(name="en_no_marks", lang="a", tokens=("none", 260)),
    dict(name="en_mixed", lang="a", tokens=("mixed", 500)),
    dict(name="en_quoted", lang="a", tokens=("quoted", 300)),
    dict(name="en_short", lang="a", tokens=("sentences", 9)),
    dict(name="es_sentences", lang="e", text="Hola mundo. " * 60 + "\n\n" + "Una frase muy larga sin puntos " * 25 + "\n \n" + "Fin!"),
    dict(name="es_no_split", lang="e", text="Buenos dias. Como estas? " * 30, split_pattern=None),
    dict(name="ja_list", lang="j", text=["Primero. Segundo! Tercero?", "", "Otro"]),
]


# ---- serving shell: one scripted scenario run through a broker module (the reference's server_inference.py or this package's)
def broker_scenario(mod):
    """Submits a fixed set of requests to ``mod.InferenceBroker`` while its worker is held inside the first request, so that the worker then sees them all
    at once: continuous sessions (two keys, one session whose step raises, one request the session refuses), fixed-window batches (key A: more requests
    than ``max_batch_size``, one cancelled while waiting; key B alone), serial requests (one raising), a request whose adapter disappears.  Returns the
    adapter / session call trace and every request's result chunks."""
    import threading

    trace, gate, started = [], threading.Event(), threading.Event()

    class Session:
        def __init__(self, key):
            self.key, self.active = key, []
            trace.append(["create", key])

        @property
        def idle(self):
            return not self.active

        def submit(self, req):
            trace.append(["submit", self.key, req.payload["id"]])
            if req.payload.get("reject"):
                raise RuntimeError("session refuses this request")
            self.active.append([req, req.payload["steps"]])

        def step(self):
            trace.append(["step", self.key, [r.payload["id"] for r, _ in self.active]])
            if any(r.payload.get("explode") for r, _ in self.active):
                raise RuntimeError("boom")
            for item in list(self.active):
                item[0].emit_data(["chunk", item[0].payload["id"], item[1]])
                item[1] -= 1
                if item[1] == 0:
                    item[0].emit_done()
                    self.active.remove(item)

        def fail(self, err):
            trace.append(["fail", self.key, str(err)])
            for r, _ in self.active:
                r.emit_error(err)
                r.emit_done()
            self.active = []

    class Adapter:
        max_batch_size = 3

        def supports_batch(self, r):
            return bool(r.payload.get("batch"))

        def batch_key(self, r):
            return r.payload.get("key")

        def supports_continuous_batch(self, r):
            return "ckey" in r.payload

        def continuous_batch_key(self, r):
            return r.payload["ckey"]

        def create_continuous_batch_session(self, r):
            return Session(r.payload["ckey"])

        def run_serial(self, r):
            trace.append(["serial", r.payload["id"]])
            if r.payload.get("gate"):
                started.set()
                gate.wait(10)
            if r.payload.get("raise"):
                raise ValueError("bad request")
            r.emit_data(["out", r.payload["id"]])
            r.emit_done()

        def run_batch(self, reqs):
            trace.append(["batch", [r.payload["id"] for r in reqs]])
            for r in reqs:
                r.emit_data(["out", r.payload["id"]])
                r.emit_done()

    broker = mod.InferenceBroker(idle_poll_s=0.01)
    broker.register_adapter("tts", Adapter())
    broker.register_adapter("gone", Adapter())
    unknown = None
    try:
        broker.submit(endpoint_kind="nope", model_name="m", payload={"id": -1})
    except ValueError as e:
        unknown = str(e)
    payloads = [dict(id=0, gate=True), dict(id=1, batch=True, key="A"), dict(id=2), dict(id=3, batch=True, key="A"), dict(id=4, batch=True, key="B"),
                dict(id=5, batch=True, key="A"), dict(id=6, batch=True, key="A"), dict(id=7, batch=True, key="A"), dict(id=8, ckey="X", steps=2),
                dict(id=9, ckey="X", steps=3), dict(id=10, ckey="Y", steps=1), dict(id=11, endpoint="gone"), dict(id=12, batch=True, key="A"),
                dict(id=13, ckey="Z", steps=2, explode=True), dict(id=14, ckey="X", steps=1, reject=True), dict(id=15, **{"raise": True}),
                dict(id=16, batch=True, key="A", model="other")]
    handles = []
    for i, pl in enumerate(payloads):
        handles.append(broker.submit(endpoint_kind=pl.get("endpoint", "tts"), model_name=pl.get("model", "m"), payload=pl))
        if i == 0:
            assert started.wait(10)
    handles[7].cancel()
    broker._adapters.pop("gone")
    gate.set()
    chunks = {}
    for pl, h in zip(payloads, handles):
        got = []
        if pl["id"] != 7:                       # the cancelled request is dropped without a word
            while True:
                c = h.result_queue.get(timeout=10)
                got.append([c.kind, c.payload if c.kind == "data" else (str(c.error) if c.kind == "error" else None)])
                if c.kind == "done":
                    break
        chunks[str(pl["id"])] = got
    broker.stop_and_join()
    return dict(unknown_endpoint=unknown, trace=trace, chunks=chunks)


# ---- Whisper decode host helpers (make_reference_fixtures.run_whisper_host and tests/test_whisper_generate_cpu.py)
WHISPER_HOST_CASES = dict(
    initial=[dict(), dict(without_timestamps=True), dict(prefix=list(range(300, 340))), dict(prompt=list(range(400, 460))),
             dict(prompt=list(range(400, 460)), prefix=[9, 8, 7], without_timestamps=True), dict(prompt="hello there", prefix=" well "),
             dict(prefix=list(range(300, 340)), sample_len=30), dict(prompt=[5], sample_len=3)],
    suppress=["-1", "1,2,-1", [5, 6], "7", None, [-1, 50300]],
    rank=[dict(tokens=[[1, 2, 3], [1, 2], [4, 5, 6, 7]], sum_logprobs=[-3.0, -2.5, -3.2], length_penalty=None),
          dict(tokens=[[1, 2, 3], [1, 2], [4, 5, 6, 7]], sum_logprobs=[-3.0, -2.5, -3.2], length_penalty=0.6),
          dict(tokens=[[1], [2]], sum_logprobs=[-1.0, -1.0], length_penalty=None),
          dict(tokens=[[1, 2, 3, 4, 5, 6], [1]], sum_logprobs=[-6.6, -1.2], length_penalty=1.0)],
    text=["hello hello hello hello hello", "The quick brown fox jumps over the lazy dog.", "a", "ĉu vi ŝatas ĝin? " * 7],
)
WHISPER_HOST_N_CTX = 64


# ---- more checkpoints in their published form for the sanitize pins (Whisper from the HF hub, Qwen3-TTS, KittenTTS exports)
_WHISPER_TO_HF = [("decoder.positional_embedding", "decoder.embed_positions.weight"), ("encoder.ln_post.", "encoder.layer_norm."), ("decoder.ln.", "decoder.layer_norm."),
                  ("encoder.blocks.", "encoder.layers."), ("decoder.blocks.", "decoder.layers."), (".cross_attn_ln.", ".encoder_attn_layer_norm."),
                  (".attn_ln.", ".self_attn_layer_norm."), (".mlp_ln.", ".final_layer_norm."), (".mlp1.", ".fc1."), (".mlp2.", ".fc2."),
                  (".cross_attn.query.", ".encoder_attn.q_proj."), (".cross_attn.key.", ".encoder_attn.k_proj."), (".cross_attn.value.", ".encoder_attn.v_proj."),
                  (".cross_attn.out.", ".encoder_attn.out_proj."), (".attn.query.", ".self_attn.q_proj."), (".attn.key.", ".self_attn.k_proj."),
                  (".attn.value.", ".self_attn.v_proj."), (".attn.out.", ".self_attn.out_proj."), ("decoder.token_embedding.", "decoder.embed_tokens.")]


def whisper_hf_checkpoint(w):
    """This package's Whisper weights (MLX names, conv (out, K, in)) under the HuggingFace names with the ``model.`` prefix, PyTorch conv layout, plus the
    sinusoid table HF checkpoints carry for the encoder (dropped by sanitize)."""
    out = {}
    for k, v in w.items():
        for mlx_name, hf_name in _WHISPER_TO_HF:
            if mlx_name in k:
                k = k.replace(mlx_name, hf_name)
        if ("conv1.weight" in k or "conv2.weight" in k) and v.dim() == 3:
            v = v.permute(0, 2, 1).contiguous()
        out["model." + k] = v
    out["model.encoder.embed_positions.weight"] = torch.zeros(4, 4)
    return out


def qwen3_model_checkpoint(seed=0):
    """Keys / shapes that walk every branch of Qwen3-TTS ``Model.sanitize`` and its layout heuristic (qwen3_tts.py:123-157, 2914-2937)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape: torch.randn(*shape, generator=g)
    return {"talker.model.layers.0.self_attn.q_proj.weight": r(8, 8), "talker.model.rotary.position_ids": torch.arange(6)[None],
            "speaker_encoder.blocks.0.conv.weight": r(64, 80, 5), "speaker_encoder.blocks.0.conv.bias": r(64), "speaker_encoder.asp.conv.weight": r(128, 1, 200),
            "speaker_encoder.blocks.1.conv.weight": r(128, 1, 7), "speech_tokenizer.decoder.pre_conv.conv.weight": r(32, 100, 1),
            "speech_tokenizer.decoder.x.conv.weight": r(32, 3, 1), "speech_tokenizer.decoder.y.conv.weight": r(16, 4, 9), "speaker_encoder.fc.weight": r(8, 20, 1),
            "speaker_encoder.fc.bias": r(8), "talker.other.weight": r(3, 9, 2), "speech_tokenizer.decoder.upconv.weight": r(6, 12, 12)}


def kitten_alpha_checkpoints():
    r = lambda n: torch.arange(n, dtype=torch.float32)
    return [{"decoder.generator.resblocks.0.alpha1.0": r(3), "decoder.generator.resblocks.0.alpha2.1": r(4), "bert.embeddings.word_embeddings.weight": r(5)},
            {"decoder.generator.resblocks.0.alpha1_0": r(3), "decoder.generator.resblocks.0.alpha2_1": r(4)},
            {"decoder.generator.resblocks.0.alpha1.0": r(3), "decoder.generator.resblocks.1.alpha1_0": r(2)},
            {"text_encoder.lstm.weight_ih_l0": r(6)}]


# ---- Qwen3-TTS continuous batching: one scripted scenario driven through a session object (the reference's Qwen3TTSBatchSession over a scripted
# model, or this package's over a scripted slot engine): who is advanced / admitted at every step, which events come out, in which order
QWEN3_SESSION = dict(
    max_batch=3, max_tokens=6, eos=2999,
    # sequence id -> first-codebook tokens it samples before EOS (0: EOS on its first frame = empty event; >= max_tokens: cut by max_tokens)
    script={1: 3, 2: 0, 3: 9, 4: 2, 5: 1, 6: 4, 7: 6, 8: 2},
    arrivals={0: [1, 2, 3, 4], 3: [5], 4: [6, 7, 8]},
    cancels={5: [8], 6: [7]},        # step -> sequence ids cancelled BEFORE that step runs (8 while still pending, 7 while active)
)


def qwen3_session_drive(session, make_item, cfg=QWEN3_SESSION, max_steps=60):
    """Feeds the scripted arrivals / cancellations to ``session`` and steps it until idle.  Returns [[step, [[sequence id, token_count, samples], ...],
    available_slots after the step, idle after the step], ...]."""
    rows, step = [], 0
    last = max(list(cfg["arrivals"]) + list(cfg["cancels"]))
    while step < max_steps and (step <= last or not session.idle):
        if step in cfg["arrivals"]:
            session.add([make_item(i) for i in cfg["arrivals"][step]])
        for sid in cfg["cancels"].get(step, []):
            session.cancel(sid)
        ev = session.step()
        rows.append([step, [[int(e.sequence_id), int(e.token_count), int(e.samples)] for e in ev], int(session.available_slots), bool(session.idle)])
        step += 1
    return rows
