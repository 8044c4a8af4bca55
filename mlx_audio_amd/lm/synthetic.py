"""Seeded synthetic parameters for the decoder-only stacks (no network => no checkpoints), canonical names
(see ``mlx_audio_amd/lm/stack.py``), bf16-representable float32 values."""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor


def make_stack_weights(cfg, seed: int = 0, gain: float = 0.5) -> Dict[str, Tensor]:
    """Seeded synthetic stack parameters under the canonical names, bf16-representable float32."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def r16(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def lin(name, n_out, n_in, bias, gn=1.0):
        w[name + ".weight"] = r16(torch.randn(n_out, n_in, generator=g) * gn / math.sqrt(n_in))
        if bias:
            w[name + ".bias"] = r16(torch.randn(n_out, generator=g) * 0.02)

    def norm(name, n):
        w[name + ".weight"] = r16(1.0 + 0.1 * torch.randn(n, generator=g))
        if cfg.norm == "layer":
            w[name + ".bias"] = r16(0.05 * torch.randn(n, generator=g))

    D, H, G, dh = cfg.d_model, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
    for i in range(cfg.n_layers):
        p = f"layers.{i}."
        norm(p + "attn_norm", D)
        lin(p + "wq", H * dh, D, cfg.attn_bias)
        lin(p + "wk", G * dh, D, cfg.attn_bias)
        lin(p + "wv", G * dh, D, cfg.attn_bias)
        lin(p + "wo", D, H * dh, cfg.attn_bias, gain)
        if cfg.qk_norm:
            w[p + "q_norm.weight"] = r16(1.0 + 0.1 * torch.randn(dh, generator=g))
            w[p + "k_norm.weight"] = r16(1.0 + 0.1 * torch.randn(dh, generator=g))
        norm(p + "mlp_norm", D)
        if cfg.mlp == "swiglu":
            lin(p + "w_gate", cfg.d_ff, D, cfg.mlp_bias)
            lin(p + "w_up", cfg.d_ff, D, cfg.mlp_bias)
            lin(p + "w_down", D, cfg.d_ff, cfg.mlp_bias, gain)
        else:
            lin(p + "w1", cfg.d_ff, D, cfg.mlp_bias)
            lin(p + "w2", D, cfg.d_ff, cfg.mlp_bias, gain)
        if cfg.layer_scale:
            w[p + "ls1"] = r16(0.5 + 0.1 * torch.randn(D, generator=g))
            w[p + "ls2"] = r16(0.5 + 0.1 * torch.randn(D, generator=g))
    if cfg.final_norm:
        norm("final_norm", D)
    return w
