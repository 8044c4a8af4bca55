"""PyTorch-CPU restatement of the Mimi codec DECODE and ENCODE paths (TEST ORACLE, not product).

Follows ``codec/models/mimi`` of the reference:
  * ``mimi.py:36-91``            mimi_202407 configuration (SEANet ratios [8,6,5,4], 64 filters, 8-layer transformer d 512 / 8 heads /
                                 ff 2048 / context 250 / LayerScale / traditional RoPE, 32 codebooks x 2048 x 256, 12.5 Hz)
  * ``mimi.py:155-176``          Mimi.decode: quantizer.decode -> upsample -> decoder_transformer -> SEANet decoder
  * ``modules/quantization.py``  EuclideanCodebook (embedding = embedding_sum / max(cluster_usage, 1e-5), :14-31),
                                 Split / ResidualVectorQuantizer.decode (:93-100, 131-137, 186-191)
  * ``modules/conv.py:181-331``  StreamableConv1d (causal: left pad (K-1)*dil, "constant"), StreamableConvTranspose1d (causal: trim
                                 K - stride on the right), ConvTrUpsample1d (depthwise, K = 2*stride, no bias)
  * ``modules/transformer.py``   via oracle.lm_ref.StackRef (LayerNorm 1e-5, fused in_proj, interleaved RoPE, causal + context window,
                                 gelu_approx MLP, LayerScale, no final norm)
  * ``modules/seanet.py:54-110, 206-300``  SeanetResnetBlock (ELU -> conv k3 -> ELU -> conv k1, true skip), DecoderLayer, SeanetDecoder
The non-streaming ``decode`` is restated; the streaming ``decode_step`` (one frame per call, CSM's usage) produces the same samples for
causal convolutions, which is what the product computes in one batch per utterance (state reset per utterance).
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files for
Mimi (imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) on a seeded tiny checkpoint, and
tests/test_reference_fixtures_cpu.py holds this oracle to the result -- ``Mimi.decode`` and frame-by-frame ``decode_step`` (30 frames, attention context 20): 1e-6 of the waveform.
The ENCODE path (``MimiEncoderRef``: mimi.py:146-153 ``Mimi.encode`` = SeanetEncoder (seanet.py:118-205: strided causal convs K = 2 * ratio) ->
encoder_transformer -> ConvDownsample1d (conv.py:333-355: K = 2 * stride, "edge" left padding, no bias) -> SplitResidualVectorQuantizer.encode
(quantization.py:37-45, 84-96, 125-128, 174-179: per layer argmin of |e|^2 / 2 - x . e, residual in float32)) is pinned the same way:
``ref_mimi_encode.npz`` holds the reference's own codes and pre-quantiser latent for a seeded clip (round 3).
The reference's own tests pin shapes / token-rule cases only (reproduced in tests/test_oracle_golden.py); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F

from .lm_ref import StackConfig, StackRef

Tensor = torch.Tensor


@dataclass
class MimiConfig:
    """mimi_202407(num_codebooks) (mimi.py:36-91) flattened to what decode needs."""
    dimension: int = 512
    nfilters: int = 64
    ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])
    ksize: int = 7
    residual_ksize: int = 3
    last_ksize: int = 3
    compress: int = 2
    num_heads: int = 8
    num_layers: int = 8
    dim_feedforward: int = 2048
    context: int = 250
    max_period: float = 10000.0
    max_seq_len: int = 8192
    quantizer_nq: int = 32
    quantizer_bins: int = 2048
    quantizer_dim: int = 256
    upsample_stride: int = 2      # encoder frame rate 25 Hz / codec frame rate 12.5 Hz
    sample_rate: int = 24000
    rope_interleaved: bool = True   # nn.RoPE(traditional=...) (transformer.py:38, 76); the Qwen3-TTS tokenizer encoder: False
    attn_window: int = -1           # -1: ``context``; 0: plain causal (speech_tokenizer.py:1046-1053 hands the transformer an explicit causal mask)


def mimi_stack_config(cfg) -> StackConfig:
    return StackConfig(d_model=cfg.dimension, n_layers=cfg.num_layers, n_heads=cfg.num_heads, n_kv_heads=cfg.num_heads,
                       head_dim=cfg.dimension // cfg.num_heads, d_ff=cfg.dim_feedforward, norm="layer", norm_eps=1e-5, rope_theta=cfg.max_period,
                       rope_interleaved=bool(getattr(cfg, "rope_interleaved", True)), max_pos=cfg.max_seq_len, attn_bias=False, mlp="gelu_tanh", mlp_bias=False,
                       layer_scale=True, causal=True, window=cfg.context if getattr(cfg, "attn_window", -1) < 0 else int(cfg.attn_window), final_norm=False)


def canonical_stack_weights(w: Dict[str, Tensor], prefix: str, cfg) -> Dict[str, Tensor]:
    """decoder_transformer.transformer.layers.N.{self_attn.in_proj, ...} -> canonical names; in_proj rows are [q | k | v] (transformer.py:88-92)."""
    d = cfg.dimension
    out = {}
    for i in range(cfg.num_layers):
        p = f"{prefix}layers.{i}."
        ip = w[p + "self_attn.in_proj.weight"]
        out[f"layers.{i}.wq.weight"], out[f"layers.{i}.wk.weight"], out[f"layers.{i}.wv.weight"] = ip[:d], ip[d:2 * d], ip[2 * d:]
        out[f"layers.{i}.wo.weight"] = w[p + "self_attn.out_proj.weight"]
        for src, dst in (("norm1", "attn_norm"), ("norm2", "mlp_norm")):
            out[f"layers.{i}.{dst}.weight"] = w[p + src + ".weight"]
            out[f"layers.{i}.{dst}.bias"] = w[p + src + ".bias"]
        out[f"layers.{i}.w1.weight"] = w[p + "gating.linear1.weight"]
        out[f"layers.{i}.w2.weight"] = w[p + "gating.linear2.weight"]
        out[f"layers.{i}.ls1"] = w[p + "layer_scale_1.scale"]
        out[f"layers.{i}.ls2"] = w[p + "layer_scale_2.scale"]
    return out


class MimiDecoderRef:
    def __init__(self, weights: Dict[str, Tensor], cfg: MimiConfig, dtype=torch.float32, param_dtype=torch.bfloat16):
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: (v.to(dtype) if ".codebook." in k else v.to(param_dtype).to(dtype)) for k, v in weights.items()}   # codebook statistics: never cast (quantization.py:26-30)
        self.stack = StackRef(canonical_stack_weights(weights, "decoder_transformer.transformer.", cfg), mimi_stack_config(cfg), dtype, param_dtype)
        self.total_upsample = cfg.upsample_stride
        for r in cfg.ratios:
            self.total_upsample *= r

    def _embedding(self, pfx: str) -> Tensor:
        usage = torch.clamp(self.w[pfx + ".cluster_usage"], min=1e-5)[:, None]
        return self.w[pfx + ".embedding_sum"] / usage

    def dequantize(self, codes: Tensor) -> Tensor:
        """[B, nq, N] -> [B, N, dimension]."""
        def rvq(pfx, cs):
            q = 0
            for i in range(cs.shape[1]):
                q = q + self._embedding(f"{pfx}.vq.layers.{i}.codebook")[cs[:, i]]
            return F.conv1d(q.transpose(1, 2), self.w[pfx + ".output_proj.weight"].permute(0, 2, 1)).transpose(1, 2)

        out = rvq("quantizer.rvq_first", codes[:, :1])
        if codes.shape[1] > 1:
            out = out + rvq("quantizer.rvq_rest", codes[:, 1:])
        return out

    def _conv(self, x: Tensor, name: str, dil: int = 1, elu: bool = False) -> Tensor:
        w = self.w[name + ".weight"]
        k = w.shape[1]
        if elu:
            x = F.elu(x)
        xp = F.pad(x.transpose(1, 2), ((k - 1) * dil, 0))
        return F.conv1d(xp, w.permute(0, 2, 1), self.w.get(name + ".bias"), dilation=dil).transpose(1, 2)

    def _convT(self, x: Tensor, name: str, stride: int, groups: int = 1, elu: bool = False) -> Tensor:
        w = self.w[name + ".weight"]  # (C_out, K, C_in / groups)
        k = w.shape[1]
        if elu:
            x = F.elu(x)
        y = F.conv_transpose1d(x.transpose(1, 2), w.permute(2, 0, 1) if groups == 1 else w.permute(0, 2, 1), self.w.get(name + ".bias"),
                               stride=stride, groups=groups)
        trim = max(k - stride, 0)
        if trim > 0:
            y = y[:, :, :-trim]
        return y.transpose(1, 2)

    def __call__(self, codes: Tensor, return_stages: bool = False):
        """codes int [B, nq, N] -> audio [B, 1, N * 1920]  (Mimi.decode, mimi.py:155-161)."""
        cfg = self.cfg
        st = {}
        h = self.dequantize(codes).to(self.dtype)
        st["dequant"] = h
        h = self._convT(h, "upsample.convtr.convtr.convtr", cfg.upsample_stride, groups=cfg.dimension)
        st["upsample"] = h
        h = self.stack(h)
        st["transformer"] = h
        x = self._conv(h, "decoder.init_conv1d.conv.conv")
        for i, ratio in enumerate(cfg.ratios):
            p = f"decoder.layers.{i}"
            x = self._convT(x, p + ".upsample.convtr.convtr", ratio, elu=True)
            r = x
            y = self._conv(x, p + ".residuals.0.block.0.conv.conv", elu=True)
            y = self._conv(y, p + ".residuals.0.block.1.conv.conv", elu=True)
            x = y + r
            st[f"layer{i}"] = x
        x = self._conv(x, "decoder.final_conv1d.conv.conv", elu=True)
        out = x.transpose(1, 2)
        return (out, st) if return_stages else out


class MimiEncoderRef:
    """``Mimi.encode`` (mimi.py:146-153): pcm [B, 1, S] -> codes int64 [B, nq, ceil(S / (prod(ratios) * stride))]."""

    def __init__(self, weights: Dict[str, Tensor], cfg: MimiConfig, dtype=torch.float32, param_dtype=torch.bfloat16):
        self.cfg = cfg
        self.dtype = dtype
        # (codebook statistics stay in the checkpoint's own values: quantization.py:26-47 never casts them)
        self.w = {k: (v.to(dtype) if ".codebook." in k else v.to(param_dtype).to(dtype)) for k, v in weights.items()}
        self.stack = StackRef(canonical_stack_weights(weights, "encoder_transformer.transformer.", cfg), mimi_stack_config(cfg), dtype, param_dtype)

    def _sconv(self, x: Tensor, name: str, stride: int = 1, dil: int = 1, elu: bool = False, pad_mode: str = "constant") -> Tensor:
        """StreamableConv1d.__call__ (conv.py:213-236), causal: x [B, L, C] -> [B, ceil(L / stride), C_out]."""
        w = self.w[name + ".weight"]  # (C_out, K, C_in)
        k = (w.shape[1] - 1) * dil + 1
        if elu:
            x = F.elu(x)
        L = x.shape[1]
        total = k - stride
        nframes = max(L + total - k, 0) / stride + 1.0
        import math
        ideal = (int(math.ceil(nframes)) - 1) * stride + k - total
        extra = max(0, ideal - L)
        xt = x.transpose(1, 2)
        xp = F.pad(xt, (total, extra), mode="replicate" if pad_mode == "edge" else "constant")
        return F.conv1d(xp, w.permute(0, 2, 1), self.w.get(name + ".bias"), stride=stride, dilation=dil).transpose(1, 2)

    def _embedding(self, pfx: str) -> Tensor:
        usage = torch.clamp(self.w[pfx + ".cluster_usage"], min=1e-5)[:, None]
        return self.w[pfx + ".embedding_sum"] / usage

    def latent(self, pcm: Tensor, return_stages: bool = False):
        """Everything in front of the quantiser: [B, 1, S] -> [B, T, dimension]."""
        cfg = self.cfg
        st = {}
        x = self._sconv(pcm.to(self.dtype).transpose(1, 2), "encoder.init_conv1d.conv.conv")
        for i, ratio in enumerate(reversed(cfg.ratios)):
            p = f"encoder.layers.{i}"
            y = self._sconv(x, p + ".residuals.0.block.0.conv.conv", elu=True)
            y = self._sconv(y, p + ".residuals.0.block.1.conv.conv", elu=True)
            x = y + x
            x = self._sconv(x, p + ".downsample.conv.conv", stride=ratio, elu=True)
            st[f"layer{i}"] = x
        x = self._sconv(x, "encoder.final_conv1d.conv.conv", elu=True)
        st["seanet"] = x
        x = self.stack(x)
        st["transformer"] = x
        x = self._sconv(x, "downsample.conv.conv.conv", stride=cfg.upsample_stride, pad_mode="edge")
        st["latent"] = x
        return (x, st) if return_stages else x

    def quantize(self, z: Tensor, return_margins: bool = False):
        """SplitResidualVectorQuantizer.encode on z [B, T, dimension] -> codes [B, nq, T] (and, for the margin rule of the tests, the gap between the
        best and the second-best score of every decision)."""
        cfg = self.cfg
        codes, margins = [], []
        for pfx, n in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", cfg.quantizer_nq - 1)):
            if n <= 0:
                continue
            r = F.conv1d(z.transpose(1, 2), self.w[pfx + ".input_proj.weight"].permute(0, 2, 1)).transpose(1, 2).to(torch.float32)
            for i in range(n):
                e = self._embedding(f"{pfx}.vq.layers.{i}.codebook").to(torch.float32)
                c2 = (e * e).sum(-1) / 2
                score = c2[None, None, :] - r @ e.t()
                top2 = torch.topk(score, 2, dim=-1, largest=False)
                idx = score.argmin(-1)   # the first minimum, like mx.argmin
                codes.append(idx)
                margins.append(top2.values[..., 1] - top2.values[..., 0])
                r = r - e[idx]
        out = torch.stack(codes, 1)
        return (out, torch.stack(margins, 1)) if return_margins else out

    def __call__(self, pcm: Tensor) -> Tensor:
        return self.quantize(self.latent(pcm))
