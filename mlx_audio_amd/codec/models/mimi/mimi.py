"""Mimi codec on MI355X: decode (RVQ codes -> 24 kHz waveform) and encode (waveform -> codes): host schedules over the HIP kernels.

ENCODE (``Mimi.encode``, mimi.py:146-153; round 3 -- what CSM's audio context / ``ref_audio`` and the Qwen3-TTS tokenizer's encoder need): SeanetEncoder
(seanet.py:118-205) -> encoder_transformer -> ConvDownsample1d -> SplitResidualVectorQuantizer.encode.  Every strided causal conv has K = 2 * stride
(conv.py: left pad K - stride, right pad up to a whole frame), so it runs as a 2-tap conv over rows regrouped ``[L / s, s * C]`` -- a free view of a
buffer whose length is a multiple of the stride (zero tail = the reference's "constant" right padding; ELU(0) = 0 keeps it zero through the prologue);
the 1-channel first conv is the flattened conv of the noise convs; the quantiser is ``mi355_rvq_encode`` (residual on chip across layers).

DECODE:

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
    rope_interleaved: bool = True    # Mimi: nn.RoPE(traditional=True); the Qwen3-TTS tokenizer's encoder builds the same modules with traditional=False
    attn_window: int = -1            # -1: the ``context`` window; 0: plain causal attention (Qwen3-TTS encode passes an explicit causal mask, speech_tokenizer.py:1046-1053)


def mimi_202407(num_codebooks: int) -> MimiConfig:
    return MimiConfig(quantizer_nq=num_codebooks)


def tiny_mimi_config() -> MimiConfig:
    return MimiConfig(dimension=128, nfilters=8, num_heads=2, num_layers=2, dim_feedforward=256, context=20, max_seq_len=512, quantizer_nq=4,
                      quantizer_bins=64, quantizer_dim=64)


def mimi_stack_config(cfg: MimiConfig) -> StackConfig:
    return StackConfig(d_model=cfg.dimension, n_layers=cfg.num_layers, n_heads=cfg.num_heads, n_kv_heads=cfg.num_heads,
                       head_dim=cfg.dimension // cfg.num_heads, d_ff=cfg.dim_feedforward, norm="layer", norm_eps=1e-5, rope_theta=cfg.max_period,
                       rope_interleaved=bool(getattr(cfg, "rope_interleaved", True)), max_pos=cfg.max_seq_len, attn_bias=False, mlp="gelu_tanh", mlp_bias=False,
                       layer_scale=True, causal=True, window=cfg.context if getattr(cfg, "attn_window", -1) < 0 else int(cfg.attn_window), final_norm=False)


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


def make_mimi_encoder_weights(cfg: MimiConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic ENCODE-side parameters under the reference's module paths (SeanetEncoder, encoder_transformer, downsample, the quantiser's
    input projections); the codebooks are the decode side's (``make_mimi_decoder_weights`` with the same seed: merge the two dicts)."""
    g = torch.Generator().manual_seed(7000 + seed)
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
    mult = 1
    conv("encoder.init_conv1d.conv.conv", cfg.nfilters, cfg.ksize, 1, gain=2.0)
    for i, ratio in enumerate(reversed(cfg.ratios)):
        dim = mult * cfg.nfilters
        p = f"encoder.layers.{i}"
        conv(p + ".residuals.0.block.0.conv.conv", dim // cfg.compress, cfg.residual_ksize, dim, gain=0.7)
        conv(p + ".residuals.0.block.1.conv.conv", dim, 1, dim // cfg.compress, gain=0.7)
        conv(p + ".downsample.conv.conv", 2 * dim, 2 * ratio, dim, gain=1.4)
        mult *= 2
    conv("encoder.final_conv1d.conv.conv", D, cfg.last_ksize, mult * cfg.nfilters, gain=1.4)
    for i in range(cfg.num_layers):
        p = f"encoder_transformer.transformer.layers.{i}."
        w[p + "self_attn.in_proj.weight"] = rnd(3 * D, D, std=1.0 / math.sqrt(D))
        w[p + "self_attn.out_proj.weight"] = rnd(D, D, std=1.0 / math.sqrt(D))
        for nm in ("norm1", "norm2"):
            w[p + nm + ".weight"] = r16(1.0 + 0.1 * torch.randn(D, generator=g))
            w[p + nm + ".bias"] = rnd(D, std=0.05)
        w[p + "gating.linear1.weight"] = rnd(cfg.dim_feedforward, D, std=1.0 / math.sqrt(D))
        w[p + "gating.linear2.weight"] = rnd(D, cfg.dim_feedforward, std=1.0 / math.sqrt(cfg.dim_feedforward))
        w[p + "layer_scale_1.scale"] = r16(0.3 + 0.05 * torch.randn(D, generator=g))
        w[p + "layer_scale_2.scale"] = r16(0.3 + 0.05 * torch.randn(D, generator=g))
    conv("downsample.conv.conv.conv", D, 2 * cfg.upsample_stride, D, bias=False, gain=1.0)
    for pfx in ("quantizer.rvq_first", "quantizer.rvq_rest"):
        conv(pfx + ".input_proj", cfg.quantizer_dim, 1, D, bias=False, gain=3.0 * math.sqrt(cfg.quantizer_dim))   # latents on the codebooks' scale
    return w


def make_pcm(batch: int, n_samples: int, seed: int = 0) -> torch.Tensor:
    """A seeded clip [B, 1, S]: a few partials + noise, amplitude ~0.3."""
    g = torch.Generator().manual_seed(9000 + seed)
    t = torch.arange(n_samples, dtype=torch.float32) / 24000.0
    f = 110.0 + 400.0 * torch.rand(batch, 4, generator=g)
    x = (torch.sin(2 * math.pi * f[:, :, None] * t[None, None, :]) * torch.tensor([0.2, 0.1, 0.05, 0.03])[None, :, None]).sum(1)
    return (x + 0.02 * torch.randn(batch, n_samples, generator=g))[:, None, :].contiguous()


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
                tabs.append(raw[c + ".embedding_sum"] / torch.clamp(raw[c + ".cluster_usage"], min=1e-5)[:, None])  # quantization.py:26-30 (checkpoint values)
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

    # ------------------------------------------------------------------ incremental decode (carried state)
    def new_stream(self, batch: int = 1) -> MimiDecodeStream:
        return MimiDecodeStream(self.stack.make_cache(), batch)

    def _conv_step(self, st: MimiDecodeStream, key: str, x, pc: PackedConv, y, *, elu: bool, dil: int = 1, res=None):
        """``_conv`` on a chunk: the conv's (K - 1) * dil left-context rows come from the stream (zeros before the first chunk = the one-shot pass's
        causal zero padding: ELU(0) = 0) and are refreshed from this chunk's input."""
        h = (pc.k - 1) * dil
        if h == 0:
            return self._conv(x, pc, y, elu=elu, dil=dil, res=res)
        B, L, C = x.shape
        hist = st.hist.get(key)
        ext = torch.empty((B, h + L, C), dtype=torch.float32, device=self.device)
        if hist is None:
            ext[:, :h].zero_()
        else:
            ext[:, :h].copy_(hist)
        ext[:, h:].copy_(x)
        ops.conv_gemm(ext, pc, y, dil=dil, pad=0, lout=L, pre_act=ACT_ELU if elu else ACT_NONE, res=res, precision=self.precision)
        st.hist[key] = ext[:, L:, :].clone()   # the last h rows
        return y

    def decode_step(self, codes: torch.Tensor, st: MimiDecodeStream, return_stages: bool = False):
        """codes int [B, nq, n] (the NEXT n frames of the stream, n >= 1) -> their audio [B, 1, n * 1920].  Concatenating the outputs of successive calls
        gives what ``__call__`` returns for the concatenated codes (same arithmetic per output row; kernels chosen by launch size may sum in a
        different order: <= 1e-6 of the peak, tests/test_mimi_gpu.py)."""
        cfg = self.cfg
        codes = codes.to(self.device, torch.int32).contiguous()
        B, Q, N = codes.shape
        assert Q == cfg.quantizer_nq and B == st.batch and N >= 1
        stages = {}
        h = self.dequantize(codes)
        stages["dequant"] = h
        s = cfg.upsample_stride
        # depthwise transposed conv, K = 2 s: output rows [t s, (t + 1) s) read frames t - 1 and t -> one frame of history
        ext = torch.zeros((B, N + 1, cfg.dimension), dtype=torch.float32, device=self.device)
        if "upsample" in st.hist:
            ext[:, :1].copy_(st.hist["upsample"])
        ext[:, 1:].copy_(h)
        u_ext = self._f(B, (N + 1) * s, cfg.dimension)
        ops.dwconv(ext, self.up_w, None, u_ext, pad=0, stride=s, transpose=True)
        st.hist["upsample"] = h[:, -1:, :].clone()
        u = u_ext[:, s:, :].contiguous()
        stages["upsample"] = u
        t = self.stack(u.clone() if return_stages else u, st.caches)   # appends the chunk's positions to the stream's KV caches
        stages["transformer"] = t
        x = self._f(B, t.shape[1], self.init_conv.cout)
        self._conv_step(st, "init", t, self.init_conv, x, elu=False)
        for i, lyr in enumerate(self.layers):
            Lin, ratio, cout = x.shape[1], lyr["ratio"], lyr["cout"]
            taps = lyr["up"].k
            assert taps == 2, "streaming decode: transposed convs with K = 2 * stride"
            ext = torch.zeros((B, Lin + 1, x.shape[2]), dtype=torch.float32, device=self.device)
            if f"up{i}" in st.hist:
                ext[:, :1].copy_(st.hist[f"up{i}"])
            ext[:, 1:].copy_(x)
            st.hist[f"up{i}"] = x[:, -1:, :].clone()
            y_ext = self._f(B, (Lin + 1) * ratio, cout)
            ops.conv_gemm(ext, lyr["up"], y_ext, pad=taps - 1, lout=Lin + 1 + taps - 1, pre_act=ACT_ELU, precision=self.precision,
                          up=dict(s=ratio, p=0, cout=cout, lout=(Lin + 1) * ratio))
            y = y_ext[:, ratio:, :]   # the first `ratio` rows belong to the previous chunk's last frame (recomputed without ITS history: dropped)
            hmid = self._f(B, Lin * ratio, lyr["c0"].cout)
            self._conv_step(st, f"c0_{i}", y, lyr["c0"], hmid, elu=True)
            self._conv_step(st, f"c1_{i}", hmid, lyr["c1"], y, elu=True, res=y)
            x = y
            stages[f"layer{i}"] = x
        out = self._f(B, x.shape[1], 1)
        self._conv_step(st, "final", x, self.final_conv, out, elu=True)
        st.frames += N
        audio = out.transpose(1, 2)
        return (audio, stages) if return_stages else audio

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


class MimiDecodeStream:
    """Carried state of an incremental decode (``Mimi.decode_step``, mimi.py:171-176; ``StreamableConv1d.step`` / ``StreamableConvTranspose1d.step``,
    modules/conv.py:245-331): the decoder transformer's KV caches, the last input frame of the depthwise upsampler and, for every causal conv of the
    SEANet decoder, the (K - 1) * dilation input rows that precede the next chunk (a transposed conv of stride s with K = 2 s taps: one row)."""

    def __init__(self, caches, batch: int):
        self.caches, self.batch = caches, batch
        self.hist: Dict[str, torch.Tensor] = {}
        self.frames = 0


class MimiStreamingDecoder:
    """``MimiStreamingDecoder`` (mimi.py:278-321): keeps the codec's decode state across calls; ``decode_frames(tokens [B, C, T] | [C, T])`` returns
    the waveform of exactly those frames.  (The reference loops ``decode_step`` frame by frame; here a call decodes its frames in one pass over the
    carried state -- every operator is causal, so the samples are the same.)"""

    def __init__(self, mimi) -> None:
        self._mimi = mimi
        self._stream = None

    def reset(self) -> None:
        self._stream = None

    def decode_frames(self, tokens: torch.Tensor) -> torch.Tensor:
        if tokens.dim() == 2:
            tokens = tokens[None]
        dec = self._mimi.decoder if hasattr(self._mimi, "decoder") else self._mimi
        if self._stream is None or self._stream.batch != tokens.shape[0]:
            self._stream = dec.new_stream(tokens.shape[0])
        return dec.decode_step(tokens, self._stream)


class MimiEncoder:
    """``Mimi.encode`` (mimi.py:146-153): pcm [B, 1, S] float -> codes int64 [B, nq, ceil(S / 1920)].  ``return_margins=True`` adds the gap between the best and
    the second-best codeword score of every decision ([B, nq, T] float32: a code whose gap is at float32 rounding level is a knife edge)."""

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: MimiConfig, device="cuda:0", precision: int = 2):
        ops.require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        dev = self.device
        # conv / linear weights enter bf16 MFMA images; the RVQ codebook statistics (embedding_sum, cluster_usage) do NOT: EuclideanCodebook.encode
        # (quantization.py:26-47) works in the checkpoint dtype -- float32 for the kyutai / Qwen3 tokenizer checkpoints -- and a 2^-9 rounding of the
        # codewords flips arg-min decisions (the rvq kernel takes float32 tables anyway)
        w = {k: (v.detach().to(torch.float32).cpu() if ".codebook." in k else v.detach().to(torch.float32).cpu().to(torch.bfloat16).to(torch.float32))
             for k, v in weights.items() if v.is_floating_point()}

        def conv(name):
            return ops.pack_conv(w[name + ".weight"], w.get(name + ".bias"), dev)

        def strided(name, stride):  # K = 2 * stride taps of C channels -> 2 taps of stride * C channels (rows regrouped)
            wt = w[name + ".weight"]
            cout, k, cin = wt.shape
            assert k == 2 * stride, (name, k, stride)
            return ops.pack_conv(wt.reshape(cout, 2, stride * cin).contiguous(), w.get(name + ".bias"), dev)

        w0 = w["encoder.init_conv1d.conv.conv.weight"]  # (nfilters, K, 1): flattened conv over the contiguous samples
        self.init_k = w0.shape[1]
        self.init_conv = ops.pack_conv(w0.reshape(w0.shape[0], 1, self.init_k).contiguous(), w.get("encoder.init_conv1d.conv.conv.bias"), dev)
        self.layers = []
        for i, ratio in enumerate(reversed(cfg.ratios)):
            p = f"encoder.layers.{i}"
            self.layers.append(dict(ratio=ratio, c0=conv(p + ".residuals.0.block.0.conv.conv"), c1=conv(p + ".residuals.0.block.1.conv.conv"),
                                    down=strided(p + ".downsample.conv.conv", ratio)))
        self.final_conv = conv("encoder.final_conv1d.conv.conv")
        self.stack = TransformerStack(canonical_stack_weights(w, "encoder_transformer.transformer.", cfg), mimi_stack_config(cfg), device=dev,
                                      precision=precision)
        self.down = strided("downsample.conv.conv.conv", cfg.upsample_stride)
        self.rvq = []
        for pfx, n in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", cfg.quantizer_nq - 1)):
            if n <= 0:
                continue
            tabs = []
            for i in range(n):
                c = f"{pfx}.vq.layers.{i}.codebook"
                tabs.append(w[c + ".embedding_sum"] / torch.clamp(w[c + ".cluster_usage"], min=1e-5)[:, None])  # quantization.py:26-30
            t = torch.stack(tabs, 0).contiguous()
            self.rvq.append((t.to(dev), t.transpose(1, 2).contiguous().to(dev), ((t * t).sum(-1) / 2).contiguous().to(dev), conv(pfx + ".input_proj"), n))

    def _z(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def latent(self, pcm: torch.Tensor, return_stages: bool = False):
        cfg, prec = self.cfg, self.precision
        x0 = pcm.to(self.device, torch.float32).reshape(pcm.shape[0], -1).contiguous()
        B, S = x0.shape
        st = {}
        up = lambda n, m: (n + m - 1) // m * m
        ratios = list(reversed(cfg.ratios))
        L = S
        x = self._z(B, up(L, ratios[0]), cfg.nfilters)   # rows past L stay zero: the right padding of the strided conv that reads this buffer
        ops.conv_gemm(x0[:, :, None], self.init_conv, x, lout=L, flat=dict(ldx=1, x_off=-(self.init_k - 1), channels=1), precision=prec)
        for i, lyr in enumerate(self.layers):
            r, C = lyr["ratio"], x.shape[2]
            h = self._z(B, x.shape[1], lyr["c0"].cout)
            ops.conv_gemm(x, lyr["c0"], h, pad=lyr["c0"].k - 1, lout=L, pre_act=ACT_ELU, precision=prec)
            ops.conv_gemm(h, lyr["c1"], x, lout=L, pre_act=ACT_ELU, res=x, precision=prec)
            Ln = x.shape[1] // r
            nxt_r = ratios[i + 1] if i + 1 < len(ratios) else 1
            y = self._z(B, up(Ln, nxt_r), lyr["down"].cout)
            ops.conv_gemm(x.view(B, Ln, r * C), lyr["down"], y, pad=1, lout=Ln, pre_act=ACT_ELU, precision=prec)   # K = 2 r, stride r as 2 taps of r rows
            x, L = y, Ln
            if return_stages:
                st[f"layer{i}"] = x[:, :L].clone()   # the next resblock updates this buffer in place
        t = self._z(B, L, cfg.dimension)
        ops.conv_gemm(x, self.final_conv, t, pad=self.final_conv.k - 1, lout=L, pre_act=ACT_ELU, precision=prec)
        st["seanet"] = t
        t = self.stack(t.clone() if return_stages else t)
        st["transformer"] = t
        # ConvDownsample1d (conv.py:333-355): K = 2 s, stride s, "edge" padding: K - s copies of the first row in front, copies of the last row up to a whole frame
        s = cfg.upsample_stride
        T2 = (L + s - 1) // s
        buf = torch.empty((B, s * (T2 + 1), cfg.dimension), dtype=torch.float32, device=self.device)
        buf[:, :s] = t[:, :1]
        buf[:, s:s + L] = t
        if s + L < buf.shape[1]:
            buf[:, s + L:] = t[:, L - 1:L]
        z = self._z(B, T2, cfg.dimension)
        ops.conv_gemm(buf.view(B, T2 + 1, s * cfg.dimension), self.down, z, pad=0, lout=T2, precision=prec)
        st["latent"] = z
        return (z, st) if return_stages else z

    def quantize(self, z: torch.Tensor, return_margins: bool = False):
        B, T, _ = z.shape
        codes, margins = [], []
        for tables, tables_t, c2, proj, n in self.rvq:
            zp = torch.empty((B, T, proj.cout), dtype=torch.float32, device=self.device)
            ops.conv_gemm(z, proj, zp, precision=self.precision)
            out = ops.rvq_encode(zp.view(B * T, proj.cout), tables, tables_t, c2, margins=return_margins)
            c, m = out if return_margins else (out, None)
            codes.append(c.view(B, T, n).permute(0, 2, 1))
            if return_margins:
                margins.append(m.view(B, T, n).permute(0, 2, 1))
        codes = torch.cat(codes, 1).to(torch.int64)
        return (codes, torch.cat(margins, 1)) if return_margins else codes

    def __call__(self, pcm: torch.Tensor, return_margins: bool = False):
        return self.quantize(self.latent(pcm), return_margins)


class Mimi:
    """Both halves behind the reference's surface (``encode`` / ``decode`` / ``sample_rate`` / ``frame_rate``, mimi.py:146-161, 184-190): the object CSM
    holds as ``_audio_tokenizer``.  Calling it decodes (the decode-only object's contract in this package)."""

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: MimiConfig, device="cuda:0", precision: int = 2):
        self.cfg = cfg
        self.decoder = MimiDecoder(weights, cfg, device=device, precision=precision)
        self.encoder = MimiEncoder(weights, cfg, device=device, precision=precision) if "encoder.init_conv1d.conv.conv.weight" in weights else None
        self.sample_rate, self.frame_rate = cfg.sample_rate, cfg.frame_rate
        if self.encoder is None:
            self.encode = None   # a checkpoint without the encoder half: callers probe ``getattr(tokenizer, "encode", None)``

    def encode(self, pcm: torch.Tensor) -> torch.Tensor:
        return self.encoder(pcm)

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        return self.decoder(codes)

    # incremental decode (mimi.py:138-144, 171-176): one carried state per object, like the reference's module-held caches
    def reset_state(self) -> None:
        self._stream = None

    def decode_step(self, codes: torch.Tensor) -> torch.Tensor:
        st = getattr(self, "_stream", None)
        if st is None or st.batch != codes.shape[0]:
            st = self._stream = self.decoder.new_stream(codes.shape[0])
        return self.decoder.decode_step(codes, st)

    def __call__(self, codes: torch.Tensor, **kw):
        return self.decoder(codes, **kw)
