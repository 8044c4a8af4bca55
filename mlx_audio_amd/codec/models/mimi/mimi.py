"""Mimi codec decode path (RVQ codes -> 24 kHz waveform) on MI355X: host schedule over the HIP kernels.

Mirrors ``Mimi.decode`` (``codec/models/mimi/mimi.py:155-161``) -- quantizer.decode -> ConvTrUpsample1d -> decoder transformer -> SEANet
decoder -- and the ``mimi_202407`` configuration (:36-91).  Kernel mapping:
  * split RVQ decode (quantization.py:93-100, 186-191): ``embed_sum`` over the materialised codebooks
    (embedding_sum / max(cluster_usage, 1e-5), quantization.py:26-30, evaluated once at load) + 1x1 projection GEMMs;
  * depthwise transposed-conv upsampler (conv.py:357-381): ``dwconv`` (transpose);
  * 8-layer transformer (transformer.py): ``lm.stack.TransformerStack`` (LayerNorm, interleaved RoPE, causal + 250-frame context window,
    gelu_approx MLP, LayerScale in the GEMM epilogues);
  * SEANet decoder (seanet.py:206-300): every ELU is the prologue of the conv that consumes it, causal convs are left-padded conv_gemms,
    transposed convs (K = 2*ratio) run polyphase with the causal right trim, the true-skip residual add is an epilogue.
The reference's streaming variant (``decode_step`` per 80 ms frame, kept state in every conv, conv.py:245-331) yields the same samples
for causal convolutions; this engine decodes a whole utterance per call and carries no state between calls (the reference resets only
when ``stream=True``, sesame.py:786-788 -- documented difference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch

from .... import ops
from ....lm.stack import StackConfig, TransformerStack
from ....ops import ACT_ELU, ACT_NONE, PackedConv


@dataclass
class MimiConfig:
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
    upsample_stride: int = 2
    sample_rate: int = 24000
    frame_rate: float = 12.5


def mimi_202407(num_codebooks: int) -> MimiConfig:
    return MimiConfig(quantizer_nq=num_codebooks)


def tiny_mimi_config() -> MimiConfig:
    return MimiConfig(dimension=128, nfilters=8, num_heads=2, num_layers=2, dim_feedforward=256, context=20, max_seq_len=512, quantizer_nq=4,
                      quantizer_bins=64, quantizer_dim=64)


def mimi_stack_config(cfg: MimiConfig) -> StackConfig:
    return StackConfig(d_model=cfg.dimension, n_layers=cfg.num_layers, n_heads=cfg.num_heads, n_kv_heads=cfg.num_heads,
                       head_dim=cfg.dimension // cfg.num_heads, d_ff=cfg.dim_feedforward, norm="layer", norm_eps=1e-5, rope_theta=cfg.max_period,
                       rope_interleaved=True, max_pos=cfg.max_seq_len, attn_bias=False, mlp="gelu_tanh", mlp_bias=False, layer_scale=True,
                       causal=True, window=cfg.context, final_norm=False)


def canonical_stack_weights(w: Dict[str, torch.Tensor], prefix: str, cfg: MimiConfig) -> Dict[str, torch.Tensor]:
    d = cfg.dimension
    out = {}
    for i in range(cfg.num_layers):
        p = f"{prefix}layers.{i}."
        ip = w[p + "self_attn.in_proj.weight"]  # rows [q | k | v] (transformer.py:88-92)
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


def make_mimi_decoder_weights(cfg: MimiConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic decode-side parameters under the reference's module paths (``load_pytorch_weights`` output, mimi.py:192-262)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def r16(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def rnd(*shape, std):
        return r16(torch.randn(*shape, generator=g) * std)

    def conv(name, cout, k, cin, bias=True, gain=1.0):
        w[name + ".weight"] = rnd(cout, k, cin, std=gain / math.sqrt(k * cin))
        if bias:
            w[name + ".bias"] = rnd(cout, std=0.02)

    D = cfg.dimension
    for pfx, n in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", cfg.quantizer_nq - 1)):
        for i in range(n):
            w[f"{pfx}.vq.layers.{i}.codebook.embedding_sum"] = r16(torch.randn(cfg.quantizer_bins, cfg.quantizer_dim, generator=g) * 3.0)
            w[f"{pfx}.vq.layers.{i}.codebook.cluster_usage"] = r16(torch.rand(cfg.quantizer_bins, generator=g) * 5.0 + 0.5)
        conv(pfx + ".output_proj", D, 1, cfg.quantizer_dim, bias=False, gain=1.0 / math.sqrt(max(n, 1)))
    conv("upsample.convtr.convtr.convtr", D, 2 * cfg.upsample_stride, 1, bias=False, gain=1.5)
    for i in range(cfg.num_layers):
        p = f"decoder_transformer.transformer.layers.{i}."
        w[p + "self_attn.in_proj.weight"] = rnd(3 * D, D, std=1.0 / math.sqrt(D))
        w[p + "self_attn.out_proj.weight"] = rnd(D, D, std=1.0 / math.sqrt(D))
        for nm in ("norm1", "norm2"):
            w[p + nm + ".weight"] = r16(1.0 + 0.1 * torch.randn(D, generator=g))
            w[p + nm + ".bias"] = rnd(D, std=0.05)
        w[p + "gating.linear1.weight"] = rnd(cfg.dim_feedforward, D, std=1.0 / math.sqrt(D))
        w[p + "gating.linear2.weight"] = rnd(D, cfg.dim_feedforward, std=1.0 / math.sqrt(cfg.dim_feedforward))
        w[p + "layer_scale_1.scale"] = r16(0.3 + 0.05 * torch.randn(D, generator=g))
        w[p + "layer_scale_2.scale"] = r16(0.3 + 0.05 * torch.randn(D, generator=g))
    mult = 1 << len(cfg.ratios)
    conv("decoder.init_conv1d.conv.conv", mult * cfg.nfilters, cfg.ksize, D)
    for i, ratio in enumerate(cfg.ratios):
        cin, cout = mult * cfg.nfilters, mult * cfg.nfilters // 2
        p = f"decoder.layers.{i}"
        conv(p + ".upsample.convtr.convtr", cout, 2 * ratio, cin, gain=math.sqrt(ratio))
        conv(p + ".residuals.0.block.0.conv.conv", cout // cfg.compress, cfg.residual_ksize, cout, gain=0.7)
        conv(p + ".residuals.0.block.1.conv.conv", cout, 1, cout // cfg.compress, gain=0.7)
        mult //= 2
    conv("decoder.final_conv1d.conv.conv", 1, cfg.last_ksize, cfg.nfilters, gain=0.3)
    return w


def make_codes(batch: int, n_frames: int, cfg: MimiConfig, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(3000 + seed)
    return torch.randint(0, cfg.quantizer_bins, (batch, cfg.quantizer_nq, n_frames), generator=g, dtype=torch.int64)


class MimiDecoder:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: MimiConfig, device="cuda:0", precision: int = 2):
        ops.require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        dev = self.device
        raw = {k: v.detach().to(torch.float32).cpu() for k, v in weights.items() if v.is_floating_point()}
        w = {k: v.to(torch.bfloat16).to(torch.float32) for k, v in raw.items()}  # checkpoint dtype bf16
        self.total_upsample = cfg.upsample_stride
        for r in cfg.ratios:
            self.total_upsample *= r

        def conv(name):
            return ops.pack_conv(w[name + ".weight"], w.get(name + ".bias"), dev)

        self.rvq = []
        for pfx, n in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", cfg.quantizer_nq - 1)):
            if n <= 0:
                continue
            tabs = []
            for i in range(n):
                c = f"{pfx}.vq.layers.{i}.codebook"
                tabs.append(w[c + ".embedding_sum"] / torch.clamp(w[c + ".cluster_usage"], min=1e-5)[:, None])  # quantization.py:26-30
            table = torch.cat(tabs, 0).contiguous().to(dev)
            offs = torch.tensor([i * cfg.quantizer_bins for i in range(n)], dtype=torch.int32, device=dev)
            self.rvq.append((table, offs, conv(pfx + ".output_proj"), n))
        self.up_w = w["upsample.convtr.convtr.convtr.weight"][:, :, 0].contiguous().to(dev)  # (C, K)
        self.stack = TransformerStack(canonical_stack_weights(w, "decoder_transformer.transformer.", cfg), mimi_stack_config(cfg), device=dev,
                                      precision=precision)
        self.init_conv = conv("decoder.init_conv1d.conv.conv")
        self.layers = []
        mult = 1 << len(cfg.ratios)
        for i, ratio in enumerate(cfg.ratios):
            p = f"decoder.layers.{i}"
            cout = mult * cfg.nfilters // 2
            self.layers.append(dict(ratio=ratio, cout=cout,
                                    up=ops.pack_conv_transpose(w[p + ".upsample.convtr.convtr.weight"], w.get(p + ".upsample.convtr.convtr.bias"), ratio, dev),
                                    c0=conv(p + ".residuals.0.block.0.conv.conv"), c1=conv(p + ".residuals.0.block.1.conv.conv")))
            mult //= 2
        self.final_conv = conv("decoder.final_conv1d.conv.conv")

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _conv(self, x, pc: PackedConv, y, *, elu: bool, dil: int = 1, res=None):
        return ops.conv_gemm(x, pc, y, dil=dil, pad=(pc.k - 1) * dil, pre_act=ACT_ELU if elu else ACT_NONE, res=res, precision=self.precision)

    def dequantize(self, codes: torch.Tensor) -> torch.Tensor:
        B, Q, N = codes.shape
        out = self._f(B, N, self.cfg.dimension)
        q0 = 0
        for gi, (table, offs, proj, n) in enumerate(self.rvq):
            ids = codes[:, q0:q0 + n, :].permute(0, 2, 1)
            summed = self._f(B, N, table.shape[1])
            ops.embed_sum(table, ids, summed, slot_offset=offs)
            ops.conv_gemm(summed, proj, out, accumulate=gi > 0, precision=self.precision)
            q0 += n
        return out

    def __call__(self, codes: torch.Tensor, return_stages: bool = False):
        """codes int [B, nq, N] -> audio [B, 1, N * 1920]."""
        cfg = self.cfg
        codes = codes.to(self.device, torch.int32).contiguous()
        B, Q, N = codes.shape
        assert Q == cfg.quantizer_nq
        st = {}
        h = self.dequantize(codes)
        st["dequant"] = h
        s = cfg.upsample_stride
        u = self._f(B, N * s, cfg.dimension)
        ops.dwconv(h, self.up_w, None, u, pad=0, stride=s, transpose=True)  # K = 2s, causal: the K - s tail rows are never produced
        st["upsample"] = u
        t = self.stack(u.clone() if return_stages else u)
        st["transformer"] = t
        x = self._f(B, t.shape[1], self.init_conv.cout)
        self._conv(t, self.init_conv, x, elu=False)
        for i, lyr in enumerate(self.layers):
            Lin, ratio, cout = x.shape[1], lyr["ratio"], lyr["cout"]
            y = self._f(B, Lin * ratio, cout)
            taps = lyr["up"].k
            ops.conv_gemm(x, lyr["up"], y, pad=taps - 1, lout=Lin + taps - 1, pre_act=ACT_ELU, precision=self.precision,
                          up=dict(s=ratio, p=0, cout=cout, lout=Lin * ratio))
            hmid = self._f(B, Lin * ratio, lyr["c0"].cout)
            self._conv(y, lyr["c0"], hmid, elu=True)
            self._conv(hmid, lyr["c1"], y, elu=True, res=y)
            x = y
            st[f"layer{i}"] = x
        out = self._f(B, x.shape[1], 1)
        self._conv(x, self.final_conv, out, elu=True)
        audio = out.transpose(1, 2)
        return (audio, st) if return_stages else audio
