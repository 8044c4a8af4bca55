"""CSM (Sesame) behind the reference's model protocol (``Model(config)``, ``sanitize``, ``generate``, ``generate_result`` --
``tts/models/sesame/sesame.py:450-866``), computing on MI355X through ``CSMEngine`` (Llama backbone + depth decoder frame loop) and the Mimi
decoder engine.

Same as the reference: constructor from the checkpoint's config dict (explicit HF-style sizes or the ``backbone_flavor`` / ``decoder_flavor``
presets, sesame.py:165-299), checkpoint key handling (``sanitize`` :577-604), the frame layout of prompts (``_tokenize_text_segment`` :496-521:
[n, 33] tokens + mask, text in the last column), the 2048-position guard, sampler defaults (temperature 0.9, top-k 50), the generator protocol
and ``GenerationResult`` fields.

Audio context (``context`` segments with audio, ``ref_audio`` + ``ref_text``) runs through the Mimi ENCODER (``codec.models.mimi.MimiEncoder``, round 3)
when the Mimi checkpoint carries the encoder half (``encoder.*``, ``encoder_transformer.*``, ``downsample.*``, the quantiser's ``input_proj``); a
decode-only checkpoint raises.  Not in this build (raise, never silently degrade): named voices / the default voice prompt (they are downloaded from
the hub); watermarking (``silentcipher`` is an optional dependency the reference also skips when it is missing, sesame.py:471-474).  The Mimi decoder is reset
per utterance (the reference's non-streaming path inherits state from the previous call, sesame.py:786-788: documented difference).
"""
from __future__ import annotations

import os
import re
import time
import warnings
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch

from ..base import GenerationResult
from .engine import CSMConfig, CSMEngine, llama_stack

_FLAVORS = {  # create_llama_model_args (sesame.py:204-299): (hidden, layers, heads, kv heads, head_dim, intermediate)
    "llama-1B": (2048, 16, 32, 8, 64, 8192),
    "llama-8B": (4096, 32, 32, 8, 128, 14336),
    "llama-100M": (1024, 4, 8, 2, 128, 8192),
    "llama-300M": (1536, 8, 24, 6, 64, 6912),
}


@dataclass
class Segment:
    speaker: int
    text: str
    audio: Optional[torch.Tensor]  # (num_samples,), 24 kHz


def csm_config_from_dict(config: Dict) -> CSMConfig:
    """Explicit sizes (``create_llama_model_args_for_backbone / _for_decoder``, sesame.py:165-201) when the config carries them, else the flavor
    presets.  RoPE: theta (default 5e5) and the Llama-3 factor from ``rope_scaling`` (no ``factor`` key = 1.0 = unscaled, as attention.py:132), 2048 positions."""
    def stack(d: Dict, flavor: Optional[str]):
        # explicit sizes need a rope_scaling dict (sesame.py:165-201 reads cfg.rope_scaling; attention.py:132 ``.get("factor", 1.0)``); without one the
        # reference's constructor fails and its caller falls back to the flavor presets
        if "hidden_size" in d and "num_hidden_layers" in d and (isinstance(d.get("rope_scaling"), dict) or flavor not in _FLAVORS):
            hd = d.get("head_dim") or d["hidden_size"] // d["num_attention_heads"]
            sc = llama_stack(d["hidden_size"], d["num_hidden_layers"], d["num_attention_heads"], d.get("num_key_value_heads", d["num_attention_heads"]),
                             hd, d["intermediate_size"])
            sc.norm_eps = float(d.get("rms_norm_eps", 1e-5))
            sc.rope_theta = float(d.get("rope_theta", 500000.0))
            sc.rope_llama3_factor = float((d.get("rope_scaling") or {}).get("factor", 1.0))   # attention.py:132: no factor = no scaling
            sc.max_pos = int(d.get("max_position_embeddings", 2048))
            return sc
        if flavor not in _FLAVORS:
            raise ValueError(f"Unknown flavor: {flavor}")
        return llama_stack(*_FLAVORS[flavor])

    return CSMConfig(backbone=stack(config, config.get("backbone_flavor")), decoder=stack(config.get("depth_decoder_config") or {}, config.get("decoder_flavor")),
                     audio_vocab_size=int(config.get("audio_vocab_size", 2051)), audio_num_codebooks=int(config.get("audio_num_codebooks", 32)),
                     text_vocab_size=int(config.get("text_vocab_size", 128256)))


_CANON = {"self_attn.q_proj": "wq", "self_attn.k_proj": "wk", "self_attn.v_proj": "wv", "self_attn.o_proj": "wo", "mlp.gate_proj": "w_gate",
          "mlp.up_proj": "w_up", "mlp.down_proj": "w_down", "input_layernorm": "attn_norm", "post_attention_layernorm": "mlp_norm"}


def _canonical(w: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Reference module paths after ``sanitize`` (``model.backbone.layers.N.self_attn.q_proj.weight`` ...) -> the engine's canonical stack names."""
    out = {}
    for k, v in w.items():
        k = k[len("model."):] if k.startswith("model.") else k
        m = re.match(r"^(backbone|decoder)\.layers\.(\d+)\.(.+)\.(weight|bias)$", k)
        if m and m.group(3) in _CANON:
            out[f"{m.group(1)}.layers.{m.group(2)}.{_CANON[m.group(3)]}.{m.group(4)}"] = v
        elif k in ("backbone.norm.weight", "decoder.norm.weight"):
            out[k.replace("norm.weight", "final_norm.weight")] = v
        else:
            out[k] = v
    return out


class Model:
    def __init__(self, config: Dict, device: str = "cuda", precision: int = 2):
        self.config = config
        self.device = device
        self.precision = precision
        self.cfg = csm_config_from_dict(config)
        self._frame_size = self.cfg.audio_num_codebooks + 1
        self._speaker_prefix_space = bool(config.get("speaker_prefix_space", False))
        self._use_default_voice_prompt = bool(config.get("use_default_voice_prompt", True))
        self._default_voice_match = bool(config.get("voice_match", True))
        self.tokenizer_repo = config.get("text_tokenizer")
        self._text_tokenizer = None   # post_load_hook / first use: AutoTokenizer on config["text_tokenizer"] or the model directory
        self._audio_tokenizer = None  # codec.models.mimi.Mimi (decode; encode when the checkpoint has the encoder half)
        self._watermarker = None
        self._sample_rate = 24000
        self.model = None             # CSMEngine, built by load_weights
        self.model_path = config.get("model_path")

    # ------------------------------------------------------------------ protocol
    def model_quant_predicate(self, p, m):
        return not p.startswith("_audio_tokenizer")

    @property
    def sample_rate(self):
        return self._sample_rate

    def eval(self):
        return self

    def sanitize(self, weights):
        """``sesame.py:577-604``: ``model.`` prefix, torchtune attention / MLP / norm names -> Llama names."""
        out = {}
        for k, v in weights.items():
            if not k.startswith("model."):
                k = "model." + k
            if "attn" in k and "self_attn" not in k:
                k = k.replace("attn", "self_attn").replace("output_proj", "o_proj")
            if "mlp" in k:
                k = k.replace("w1", "gate_proj").replace("w2", "down_proj").replace("w3", "up_proj")
            if "sa_norm" in k or "mlp_norm" in k:
                k = k.replace("sa_norm", "input_layernorm").replace("scale", "weight")
                k = k.replace("mlp_norm", "post_attention_layernorm").replace("scale", "weight")
            if "decoder.norm" in k or "backbone.norm" in k:
                k = k.replace("scale", "weight")
            out[k] = v
        return out

    def load_weights(self, weights, strict: bool = True):
        w = _canonical(dict(weights))
        try:
            self.model = CSMEngine(w, self.cfg, device=self.device, precision=self.precision)
        except KeyError as e:
            raise ValueError(f"CSM checkpoint is missing parameter {e}") from e
        return self

    @classmethod
    def post_load_hook(cls, model: "Model", model_path) -> "Model":
        """Text tokenizer from ``config["text_tokenizer"]`` (or the model directory) and the Mimi decoder from ``<model_path>/mimi`` (the
        reference pulls both from the hub in its constructor, sesame.py:462-469; there is no network here)."""
        model_path = Path(model_path)
        if model._text_tokenizer is None:
            try:
                from transformers import AutoTokenizer

                model._text_tokenizer = AutoTokenizer.from_pretrained(str(model.tokenizer_repo or model_path))
            except Exception as e:
                print(f"Warning: Could not load tokenizer: {e}")
        mimi_dir = Path(model.config.get("audio_tokenizer_path") or model_path / "mimi")
        if model._audio_tokenizer is None and mimi_dir.exists():
            from safetensors.torch import load_file

            from ....codec.models.mimi.mimi import Mimi, MimiConfig, mimi_202407

            w: Dict[str, torch.Tensor] = {}
            for f in sorted(mimi_dir.glob("*.safetensors")):
                w.update(load_file(str(f)))
            mcfg = model.config.get("audio_tokenizer_config")  # optional explicit sizes; the reference always loads mimi_202407(32) (mimi.py:36-91)
            mcfg = MimiConfig(**mcfg) if mcfg else mimi_202407(model.cfg.audio_num_codebooks)
            # both halves when the checkpoint carries the encoder (audio context / ref_audio need ``encode``); calling the object decodes
            model._audio_tokenizer = Mimi(w, mcfg, device=model.device, precision=model.precision)
        return model

    # ------------------------------------------------------------------ prompt frames
    def _tokenize_text_segment(self, text: str, speaker: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """``sesame.py:496-521``: (tokens int32 [n, 33], mask bool [n, 33]) with the text ids in the last column."""
        if self._text_tokenizer is None:
            raise ValueError("Text tokenizer not loaded (config['text_tokenizer'] or a tokenizer in the model directory)")
        prompt_text = text.lstrip() if self._speaker_prefix_space else text
        prefix = f"[{speaker}] " if self._speaker_prefix_space else f"[{speaker}]"
        ids = list(self._text_tokenizer.encode(f"{prefix}{prompt_text}"))
        frame = torch.zeros((len(ids), self._frame_size), dtype=torch.int32)
        mask = torch.zeros((len(ids), self._frame_size), dtype=torch.bool)
        frame[:, -1] = torch.tensor(ids, dtype=torch.int32)
        mask[:, -1] = True
        return frame, mask

    def _tokenize_audio(self, audio, add_eos: bool = True):
        """``sesame.py:527-559``: codes (K, T) of the audio tokenizer's ``encode`` -> (tokens int32 [T (+1), 33], mask) with the codes in the first K
        columns and, with ``add_eos``, one all-zero frame appended.  ``_audio_tokenizer`` is any object with the reference's contract
        ``encode(audio[1, 1, samples]) -> [1, K, T]`` (``codec.models.mimi.Mimi`` when the checkpoint has the encoder half)."""
        enc = getattr(self._audio_tokenizer, "encode", None)
        if enc is None:
            raise NotImplementedError("audio context needs the Mimi encoder: the loaded Mimi checkpoint has no encoder half (encoder.* / encoder_transformer.* / downsample.*)")
        codes = torch.as_tensor(enc(audio[None, None, ...]))[0].to(torch.int32).cpu()
        K = self._frame_size - 1
        if codes.shape[0] != K:
            raise ValueError(f"Audio tokenizer returned {codes.shape[0]} codebooks, expected {K}")
        if add_eos:
            codes = torch.cat([codes, torch.zeros((K, 1), dtype=codes.dtype)], dim=1)
        frame = torch.zeros((codes.shape[1], self._frame_size), dtype=torch.int32)
        mask = torch.zeros((codes.shape[1], self._frame_size), dtype=torch.bool)
        frame[:, :-1] = codes.t()
        mask[:, :-1] = True
        return frame, mask

    def _tokenize_segment(self, segment: Segment, add_eos: bool = True):
        """``sesame.py:561-575``; a segment without audio (an extension: the reference's segments always carry audio) contributes its text frames only."""
        t, tm = self._tokenize_text_segment(segment.text, segment.speaker)
        if segment.audio is None:
            return t, tm
        a, am = self._tokenize_audio(segment.audio, add_eos=add_eos)
        return torch.cat([t, a], 0), torch.cat([tm, am], 0)

    # ------------------------------------------------------------------ decode
    def _decode_frames(self, frames: torch.Tensor) -> torch.Tensor:
        """frames int [n, n_cb] -> waveform [samples] through the Mimi decoder (one causal pass over the whole utterance = the reference's
        streaming decoder fed chunk by chunk, codec/models/mimi/modules/conv.py:245-331)."""
        if self._audio_tokenizer is None:
            raise ValueError("Mimi decoder not loaded (expected <model_path>/mimi/*.safetensors or config['audio_tokenizer_path'])")
        return self._audio_tokenizer(frames.t()[None].contiguous())[0, 0]

    def generate_result(self, samples, start_time: float, stream: bool = False, audio: Optional[torch.Tensor] = None) -> GenerationResult:
        """``sesame.py:653-728``.  ``samples``: list of [1, n_cb] frames or an int tensor [n, n_cb]."""
        frames = torch.cat(list(samples), 0) if isinstance(samples, (list, tuple)) else samples
        token_count = int(frames.shape[0])
        if audio is None:
            audio = self._decode_frames(frames)
        torch.cuda.synchronize()
        seconds = time.perf_counter() - start_time
        n = int(audio.shape[0])
        assert n > 0, "No audio generated"
        dur = n / 24000
        return GenerationResult(
            audio=audio, samples=n, sample_rate=24000, segment_idx=0, token_count=token_count,
            audio_duration=f"{int(dur // 3600):02d}:{int(dur // 60):02d}:{int(dur % 60):02d}.{int((dur % 1) * 1000):03d}",
            real_time_factor=round(seconds / dur, 2) if dur > 0 else 0,
            prompt={"tokens": token_count, "tokens-per-sec": round(token_count / seconds, 2) if seconds > 0 else 0},
            audio_samples={"samples": n, "samples-per-sec": round(n / seconds, 2) if seconds > 0 else 0},
            processing_time_seconds=seconds, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9, is_streaming_chunk=stream)

    # ------------------------------------------------------------------ generate
    def generate(self, text: Union[List[str], str], voice: Optional[str] = None, speaker: int = 0, context: Optional[List[Segment]] = None,
                 split_pattern: Optional[str] = r"\n+", sampler: Optional[Callable] = None, max_audio_length_ms: float = 90_000, ref_audio=None,
                 ref_text: Optional[str] = None, stream: bool = False, streaming_interval: float = 0.5, voice_match: Optional[bool] = None, **kwargs):
        """``sesame.py:730-866``: one ``GenerationResult`` per text segment (``stream=True``: one per ``streaming_interval`` seconds of frames).
        ``temperature`` / ``top_k`` keyword arguments replace the reference's ``sampler`` callable (default 0.9 / 50 = ``make_sampler(0.9, top_k=50)``);
        ``seed``, ``gumbel`` and ``forced`` pass through to the engine for deterministic runs."""
        if self.model is None:
            raise RuntimeError("Model has no weights: call load_weights() (or mlx_audio_amd.tts.utils.load_model)")
        if sampler is not None:
            raise NotImplementedError("custom sampler callables run on the host; pass temperature= / top_k= instead")
        context = list(context or [])
        has_encoder = getattr(self._audio_tokenizer, "encode", None) is not None
        if ref_audio is not None and not has_encoder or any(s.audio is not None for s in context) and not has_encoder:
            raise NotImplementedError("voice prompts / reference audio need the Mimi encoder: the loaded Mimi checkpoint has no encoder half")
        if ref_audio is not None and isinstance(ref_audio, (str, os.PathLike)):
            from ....utils import load_audio

            ref_audio = load_audio(ref_audio, sample_rate=self.sample_rate)
        if not context and ref_audio is not None and ref_text is not None:   # sesame.py:753-755: the reference clip is the first segment
            context = [Segment(speaker=speaker, text=ref_text, audio=ref_audio)]
        elif ref_audio is None and not context and self._use_default_voice_prompt:
            if voice is not None:
                raise NotImplementedError("named voice prompts are downloaded from the hub and need the Mimi encoder; pass context= / ref_audio=")
            warnings.warn("CSM: the default voice prompt needs the hub and the Mimi encoder; generating without a voice prompt", stacklevel=2)
        if voice_match is None:
            voice_match = self._default_voice_match
        temperature, top_k = float(kwargs.get("temperature", 0.9)), int(kwargs.get("top_k", 50))
        max_audio_frames = int(max_audio_length_ms / 80)
        interval = max(1, int(streaming_interval * 12.5))
        if isinstance(text, str):
            text = re.split(split_pattern, text.strip()) if split_pattern else [text]
        gen = None
        if temperature > 0 and "gumbel" not in kwargs:
            gen = torch.Generator(device=self.model.device)
            gen.manual_seed(int(kwargs["seed"]) if kwargs.get("seed") is not None else int(torch.seed() % (2 ** 31)))
        for prompt in text:
            t0 = time.perf_counter()
            current = list(context)
            if voice_match and current:   # sesame.py:776-783: the text continues the FIRST segment's text, its audio is left open (no EOS frame)
                current = [Segment(speaker=speaker, text=(context[0].text + " " + prompt).strip(), audio=context[0].audio)]
            toks, masks = [], []
            for seg in current:
                st, sm = self._tokenize_segment(seg, add_eos=not voice_match)
                toks.append(st)
                masks.append(sm)
            if not voice_match or not current:
                gt, gm = self._tokenize_text_segment(prompt, speaker)
                toks.append(gt)
                masks.append(gm)
            prompt_tokens, prompt_mask = torch.cat(toks, 0), torch.cat(masks, 0)
            max_seq_len = 2048 - max_audio_frames
            if prompt_tokens.shape[0] >= max_seq_len:
                raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_seq_len}")
            if stream:
                # sesame.py:825-860: a partial result every ``streaming_interval`` seconds of frames, decoded by the Mimi streaming decoder (carried conv
                # / transformer state: codec/models/mimi/mimi.py:171-176, 278-321) and yielded WHILE the frame loop runs
                if self._audio_tokenizer is None:
                    raise ValueError("Mimi decoder not loaded (expected <model_path>/mimi/*.safetensors or config['audio_tokenizer_path'])")
                from ....codec.models.mimi.mimi import MimiStreamingDecoder

                sd = MimiStreamingDecoder(self._audio_tokenizer)
                for blk in self.model.generate_chunks(prompt_tokens[None], prompt_mask[None], max_audio_frames, chunk=interval, temperature=temperature,
                                                      top_k=top_k, gumbel=kwargs.get("gumbel"), forced=kwargs.get("forced"), generator=gen):
                    fr = blk[0]
                    audio = sd.decode_frames(fr.t()[None].contiguous())[0, 0]
                    yield self.generate_result(fr, t0, stream=True, audio=audio)
                    t0 = time.perf_counter()
                continue
            out = self.model.generate(prompt_tokens[None], prompt_mask[None], max_audio_frames, temperature=temperature, top_k=top_k,
                                      gumbel=kwargs.get("gumbel"), forced=kwargs.get("forced"), generator=gen)
            frames = out["frames"][0]
            if frames.shape[0] == 0:
                continue
            yield self.generate_result(frames, t0)
