"""Seeded synthetic Whisper checkpoints (no network => no real weights).

Parameter names and shapes are the reference's post-``sanitize`` MLX names (``stt/models/whisper/whisper.py:338-498``:
conv weights ``(C_out, K, C_in)``, ``key`` projections without bias, tied token embedding), so a real
``whisper-small`` fp16 checkpoint and a synthetic one are interchangeable (loader, engine, oracle, benchmark).
Values are variance-preserving random draws rounded to fp16, the checkpoint dtype of Whisper in the reference
(``Model(dims, dtype=mx.float16)``), returned as float32 tensors holding fp16-representable values.
"""
from __future__ import annotations

import math
from dataclasses import asdict, dataclass
from typing import Dict

import torch


@dataclass
class ModelDimensions:
    """``whisper.py:280-291``; defaults = whisper-small (dims pinned by ``stt/tests/test_models.py:43-54``)."""
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 768
    n_audio_head: int = 12
    n_audio_layer: int = 12
    n_vocab: int = 51865
    n_text_ctx: int = 448
    n_text_state: int = 768
    n_text_head: int = 12
    n_text_layer: int = 12

    @classmethod
    def from_dict(cls, config: dict) -> "ModelDimensions":
        """whisper.py:293-321: accepts both the MLX and the HuggingFace config spelling."""
        config = dict(config)
        if "d_model" in config or "encoder_layers" in config:
            return cls(n_mels=config.get("num_mel_bins", 128), n_audio_ctx=config.get("max_source_positions", 1500),
                       n_audio_state=config.get("d_model", 1280), n_audio_head=config.get("encoder_attention_heads", 20),
                       n_audio_layer=config.get("encoder_layers", 32), n_vocab=config.get("vocab_size", 51866),
                       n_text_ctx=config.get("max_target_positions", 448), n_text_state=config.get("d_model", 1280),
                       n_text_head=config.get("decoder_attention_heads", 20), n_text_layer=config.get("decoder_layers", 32))
        known = set(cls.__dataclass_fields__)
        return cls(**{k: v for k, v in config.items() if k in known})

    def to_dict(self) -> dict:
        return asdict(self)


WHISPER_SMALL = ModelDimensions()


def tiny_dims() -> ModelDimensions:
    """Structurally identical (head dim 64, ragged vocab, conv stem, cross attention) but small: fast parity tests."""
    return ModelDimensions(n_mels=80, n_audio_ctx=150, n_audio_state=128, n_audio_head=2, n_audio_layer=2, n_vocab=51865,
                           n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)


def _f16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float16).to(torch.float32)


def make_whisper_weights(dims: ModelDimensions = WHISPER_SMALL, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def rnd(*shape, std):
        return _f16(torch.randn(*shape, generator=g) * std)

    def linear(name, n_out, n_in, bias=True, gain=1.0):
        w[name + ".weight"] = rnd(n_out, n_in, std=gain / math.sqrt(n_in))
        if bias:
            w[name + ".bias"] = rnd(n_out, std=0.02)

    def ln(name, n):
        w[name + ".weight"] = _f16(1.0 + 0.1 * torch.randn(n, generator=g))
        w[name + ".bias"] = rnd(n, std=0.05)

    def block(pfx, n, cross):
        # decoder branches are weighted up so that the residual stream is dominated by attention / MLP outputs rather than
        # by the (tied, large) input embedding -- otherwise a random tied-embedding decoder just repeats its last input token
        gain = 3.0 if cross else 0.5
        for a in (["attn", "cross_attn"] if cross else ["attn"]):
            linear(f"{pfx}.{a}.query", n, n)
            linear(f"{pfx}.{a}.key", n, n, bias=False)
            linear(f"{pfx}.{a}.value", n, n)
            linear(f"{pfx}.{a}.out", n, n, gain=gain)
            ln(f"{pfx}.{a}_ln", n)
        linear(f"{pfx}.mlp1", 4 * n, n)
        linear(f"{pfx}.mlp2", n, 4 * n, gain=gain)
        ln(f"{pfx}.mlp_ln", n)

    na, nt = dims.n_audio_state, dims.n_text_state
    w["encoder.conv1.weight"] = rnd(na, 3, dims.n_mels, std=1.0 / math.sqrt(3 * dims.n_mels))
    w["encoder.conv1.bias"] = rnd(na, std=0.02)
    w["encoder.conv2.weight"] = rnd(na, 3, na, std=1.0 / math.sqrt(3 * na))
    w["encoder.conv2.bias"] = rnd(na, std=0.02)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", na, cross=False)
    ln("encoder.ln_post", na)
    # tied embedding / logits matrix: scaled so that logits have a standard deviation of ~5, i.e. a peaked next-token
    # distribution like a trained model's (with near-uniform logits the 1501 timestamp tokens always out-weigh the best text
    # token and ApplyTimestampRules degenerates to "everything masked")
    w["decoder.token_embedding.weight"] = rnd(dims.n_vocab, nt, std=5.0 / math.sqrt(nt))
    w["decoder.positional_embedding"] = rnd(dims.n_text_ctx, nt, std=0.02)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", nt, cross=True)
    ln("decoder.ln", nt)
    return w


def make_mel(batch: int, seed: int = 0, n_frames: int = 3000, n_mels: int = 80) -> torch.Tensor:
    """A log-mel-like input in Whisper's normalised range ((log10 + 4) / 4 lands in about [-1, 1.5])."""
    g = torch.Generator().manual_seed(1000 + seed)
    return (torch.randn(batch, n_frames, n_mels, generator=g) * 0.4).clamp_(-1.0, 1.5)
