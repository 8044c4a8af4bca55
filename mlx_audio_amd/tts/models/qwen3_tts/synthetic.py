"""Seeded synthetic Qwen3-TTS parameters (no network => no checkpoints): speech-tokenizer decoder, talker, code predictor.

Names are the reference's module paths after ``sanitize`` (speech_tokenizer.py:1220-1449 for the codec: conv weights
(C_out, K, C_in/groups), transposed-conv weights (C_out, K, C_in), codebooks already materialised as ``codebook.embed.weight``;
talker.py:825-837 for the talker: ``talker.`` prefix stripped), values bf16-representable float32.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import Qwen3TTSTalkerConfig, Qwen3TTSTokenizerDecoderConfig


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def tiny_codec_config() -> Qwen3TTSTokenizerDecoderConfig:
    """Structurally identical to the real decoder (split RVQ, transformer with LayerScale, two ConvNeXt upsamplers, four SnakeBeta
    blocks with dilations 1/3/9) at a fraction of the width."""
    return Qwen3TTSTokenizerDecoderConfig(latent_dim=128, codebook_dim=64, codebook_size=64, decoder_dim=192, hidden_size=128,
                                          intermediate_size=256, head_dim=64, num_attention_heads=2, num_hidden_layers=2,
                                          num_key_value_heads=2, num_quantizers=4, max_position_embeddings=512)


def make_codec_decoder_weights(cfg: Qwen3TTSTokenizerDecoderConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def rnd(*shape, std):
        return _r16(torch.randn(*shape, generator=g) * std)

    def conv(name, cout, k, cin, bias=True, gain=1.0):
        w[name + ".weight"] = rnd(cout, k, cin, std=gain / math.sqrt(k * cin))
        if bias:
            w[name + ".bias"] = rnd(cout, std=0.02)

    def lin(name, n_out, n_in, bias=True, gain=1.0):
        w[name + ".weight"] = rnd(n_out, n_in, std=gain / math.sqrt(n_in))
        if bias:
            w[name + ".bias"] = rnd(n_out, std=0.02)

    def snake(name, c):
        w[name + ".alpha"] = rnd(c, std=0.3)
        w[name + ".beta"] = rnd(c, std=0.3)

    vq_dim = cfg.codebook_dim // 2
    ns = cfg.num_semantic_quantizers
    for pfx, n in (("quantizer.rvq_first", ns), ("quantizer.rvq_rest", cfg.num_quantizers - ns)):
        for i in range(n):
            w[f"{pfx}.vq.layers.{i}.codebook.embed.weight"] = rnd(cfg.codebook_size, vq_dim, std=1.0 / math.sqrt(max(n, 1)))
        conv(pfx + ".output_proj", cfg.codebook_dim, 1, vq_dim, bias=False)
    conv("pre_conv.conv", cfg.latent_dim, 3, cfg.codebook_dim)
    D, H, G, dh = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    lin("pre_transformer.input_proj", D, cfg.latent_dim)
    lin("pre_transformer.output_proj", cfg.latent_dim, D)
    for i in range(cfg.num_hidden_layers):
        p = f"pre_transformer.layers.{i}."
        lin(p + "self_attn.q_proj", H * dh, D, cfg.attention_bias)
        lin(p + "self_attn.k_proj", G * dh, D, cfg.attention_bias)
        lin(p + "self_attn.v_proj", G * dh, D, cfg.attention_bias)
        lin(p + "self_attn.o_proj", D, H * dh, cfg.attention_bias)
        lin(p + "mlp.gate_proj", cfg.intermediate_size, D, False)
        lin(p + "mlp.up_proj", cfg.intermediate_size, D, False)
        lin(p + "mlp.down_proj", D, cfg.intermediate_size, False)
        w[p + "input_layernorm.weight"] = _r16(1.0 + 0.1 * torch.randn(D, generator=g))
        w[p + "post_attention_layernorm.weight"] = _r16(1.0 + 0.1 * torch.randn(D, generator=g))
        w[p + "self_attn_layer_scale.scale"] = _r16(0.3 + 0.05 * torch.randn(D, generator=g))
        w[p + "mlp_layer_scale.scale"] = _r16(0.3 + 0.05 * torch.randn(D, generator=g))
    w["pre_transformer.norm.weight"] = _r16(1.0 + 0.1 * torch.randn(D, generator=g))
    L = cfg.latent_dim
    for i, f in enumerate(cfg.upsampling_ratios):
        conv(f"upsample.{i}.0.conv", L, f, L)  # ConvTranspose1d weight (C_out, K, C_in), K == stride
        p = f"upsample.{i}.1"
        conv(p + ".dwconv.conv", L, 7, 1)
        w[p + ".norm.weight"] = _r16(1.0 + 0.1 * torch.randn(L, generator=g))
        w[p + ".norm.bias"] = rnd(L, std=0.05)
        lin(p + ".pwconv1", 4 * L, L)
        lin(p + ".pwconv2", L, 4 * L)
        w[p + ".gamma"] = _r16(0.3 + 0.05 * torch.randn(L, generator=g))
    conv("decoder.0.conv", cfg.decoder_dim, 7, L)
    for bi, rate in enumerate(cfg.upsample_rates):
        cin, cout = cfg.decoder_dim // (2 ** bi), cfg.decoder_dim // (2 ** (bi + 1))
        p = f"decoder.{bi + 1}.block"
        snake(p + ".0", cin)
        conv(p + ".1.conv", cout, 2 * rate, cin, gain=math.sqrt(rate))  # transposed: each output sees K / stride = 2 taps
        for ui in range(3):
            u = f"{p}.{ui + 2}"
            snake(u + ".act1", cout)
            conv(u + ".conv1.conv", cout, 7, cout, gain=0.4)
            snake(u + ".act2", cout)
            conv(u + ".conv2.conv", cout, 1, cout, gain=0.4)
    n = len(cfg.upsample_rates)
    cl = cfg.decoder_dim // (2 ** n)
    snake(f"decoder.{n + 1}", cl)
    conv(f"decoder.{n + 2}.conv", 1, 7, cl, gain=0.08)  # keeps most samples inside the +-1 clip
    return w


def make_codes(batch: int, n_frames: int, cfg: Qwen3TTSTokenizerDecoderConfig, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(2000 + seed)
    return torch.randint(0, cfg.codebook_size, (batch, cfg.num_quantizers, n_frames), generator=g, dtype=torch.int64)
