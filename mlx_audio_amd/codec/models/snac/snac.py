"""SNAC (multi-scale neural audio codec; waveform -> codes -> waveform) on MI355X: host schedule over the HIP kernels.

Mirrors ``mlx_audio/codec/models/snac/snac.py`` + ``layers.py`` + ``vq.py`` (constructor arguments, ``preprocess``, ``encode``, ``decode``, ``__call__``,
``decode_stream``, ``quantizer(z)``, ``quantizer.from_codes``), with the reference's op-by-op graph collapsed into:
  * ``ResidualVectorQuantize.from_codes`` (vq.py:116-137): ``out_proj(codebook[code])`` is folded once at load into one
    ``[n_codebooks * codebook_size, latent_dim]`` table; the coarse levels' ``repeat_interleave(stride)`` becomes an index map, so a frame is ONE
    ``embed_sum`` launch (rows summed in codebook order like the reference's running sum);
  * every ``Snake1d`` (layers.py:123-129, 298-306) is the PROLOGUE of the conv that consumes it (alpha and 1 / (alpha + 1e-9) precomputed);
  * ``WNConv1d`` / ``WNConvTranspose1d`` (layers.py:18-120): weight norm folded at load; dense convs are implicit GEMMs (``conv_gemm``), the
    transposed convs (K = 2 stride) run polyphase as 2-tap stride-1 GEMMs with a strided store; the depthwise k7 convs of the input stage and of
    every ``ResidualUnit`` (groups = channels, dilation 1 / 3 / 9) run on ``mi355_dwconv`` with the Snake prologue; residual adds and the final
    ``tanh`` are epilogues;
  * ``NoiseBlock`` (layers.py:256-267): ``x + noise * linear(x)`` with the Gaussian noise either supplied (parity tests) or drawn on the device.
    Reference quirk preserved: the block unpacks ``B, C, T = x.shape`` from a channels-LAST tensor, so its ``mx.random.normal((B, 1, T))`` is one
    draw per CHANNEL, constant over time (the PyTorch original draws one per time step) -- found by running the reference's own file
    (tests/golden/make_reference_fixtures.py); ``noises[i]`` is therefore ``[B, 1, channels_i]``.
Reference quirk preserved: ``WNConvTranspose1d`` hands ``groups = 1`` to MLX's ``output_padding`` slot (positional order), so each transposed
conv emits one extra sample: the reference's test pins 59 / 118 / 236 code frames -> 120 907 samples (codec/tests/test_snac.py:24-34).

``attn_window_size`` = ``None`` is the 24 kHz model (the one Orpheus-style TTS uses).  The 32 / 44 kHz models' ``LocalMHA`` (windowed attention between the
input convs and the first decoder block, attention.py:5-53) is LayerNorm + two GEMMs around one attention launch whose batch items are the windows -- built
to what the module MEANS: the reference's own transcription expects [B, C, T] data but receives channels-last [B, T, C] and raises (recorded in
tests/golden/ref_snac_local_mha_probe.json), so this path has no reference output and is held to the oracle's restatement only (PARITY UNPINNED).

Encode side (round 5; layers.py:132-156, 236-253, vq.py:10-113, snac.py:88-102), from the same kernels:
  * ``Encoder``: the 1 -> d_model k7 conv runs FLATTENED over the contiguous samples; an ``EncoderBlock`` is its three ``ResidualUnit``s (depthwise k7
    on ``mi355_dwconv`` with the Snake prologue, or dense) updating the block's activation in place inside a zeroed staging buffer, then the strided
    ``WNConv1d(K = 2 s, stride s, padding ceil(s / 2))`` as a TWO-tap conv over that buffer's rows regrouped ``[rows / s, s * C]`` (Snake(0) = 0 keeps
    the padding rows zero through the prologue); [LocalMHA]; the final k7 conv (depthwise or dense, no Snake in front);
  * ``ResidualVectorQuantize.__call__``: per level the ``stride``-frame average pool and ``in_proj`` are ONE 1-tap conv over rows regrouped
    ``[T / s, s * D]`` (weights ``W / s`` tiled s times: the same linear map), then the nearest L2-normalised codeword (``mi355_rvq_encode``), then the
    residual update ``residual -= repeat_interleave(out_proj(codebook[idx]), s)`` as one ``embed_sum`` over the NEGATED folded table with the
    level's ids repeated s times; ``z_q`` is ``from_codes`` of the result.
Weights: float32 checkpoints are held as fp16 MFMA images, activations split fp16 hi + lo (``precision = 4``); deviation from the float32 oracle
asserted in ``tests/test_snac_gpu.py`` / ``tests/test_codec_encode_gpu.py``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .... import ops
from ....ops import ACT_NONE, ACT_SNAKE, ACT_TANH, PackedConv, round_up


def make_snac_weights(latent_dim: int, decoder_dim: int, decoder_rates: List[int], vq_strides: List[int], codebook_size: int, codebook_dim: int,
                      noise: bool = True, depthwise: bool = True, seed: int = 0, attn: bool = False) -> Dict[str, torch.Tensor]:
    """Random float32 decode-side parameters of the shapes ``SNAC(...)`` allocates (reference module paths, MLX layouts)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin_g, fan_in, transpose=False, gain=1.0, bias=True):
        scale = math.sqrt(1 / fan_in)
        shape = (cin_g, k, cout) if transpose else (cout, k, cin_g)   # transposed convs are stored (in, K, out) (layers.py:92-96)
        v0 = (torch.rand(shape, generator=g) * 2 - 1) * scale * gain
        gw = torch.sqrt((v0 ** 2).sum(dim=(1, 2), keepdim=True))
        w[name + ".weight_g"] = gw * (1.0 + 0.1 * torch.randn(gw.shape, generator=g))
        w[name + ".weight_v"] = v0 / (gw + 1e-12)
        if bias:
            w[name + ".bias"] = 0.02 * torch.randn(cout, generator=g)

    def alpha(name, c):
        w[name + ".alpha"] = (1.0 + 0.3 * torch.randn(1, c, 1, generator=g)).abs() + 0.05

    n_cb = len(vq_strides)
    for i in range(n_cb):
        p = f"quantizer.quantizers.{i}."
        w[p + "codebook.weight"] = torch.randn(codebook_size, codebook_dim, generator=g)
        conv(p + "out_proj", latent_dim, 1, codebook_dim, codebook_dim, gain=1.0 / math.sqrt(n_cb))
    m = "decoder.model.layers."
    if depthwise:
        conv(m + "0", latent_dim, 7, 1, 7, gain=1.7)
        conv(m + "1", decoder_dim, 1, latent_dim, latent_dim, gain=1.7)
        nxt = 2
    else:
        conv(m + "0", decoder_dim, 7, latent_dim, 7 * latent_dim, gain=1.7)
        nxt = 1
    if attn:  # LocalMHA (attention.py:5-18): LayerNorm, to_qkv / to_out without bias
        a = f"{m}{nxt}."
        w[a + "norm.weight"] = 1.0 + 0.1 * torch.randn(decoder_dim, generator=g)
        w[a + "norm.bias"] = 0.05 * torch.randn(decoder_dim, generator=g)
        w[a + "to_qkv.weight"] = (torch.rand(3 * decoder_dim, decoder_dim, generator=g) * 2 - 1) * math.sqrt(3.0 / decoder_dim)
        w[a + "to_out.weight"] = (torch.rand(decoder_dim, decoder_dim, generator=g) * 2 - 1) * math.sqrt(1.5 / decoder_dim)
        nxt += 1
    out_dim = decoder_dim
    for i, s in enumerate(decoder_rates):
        in_dim, out_dim = decoder_dim // 2 ** i, decoder_dim // 2 ** (i + 1)
        p = f"{m}{nxt + i}.block.layers."
        alpha(p + "0", in_dim)
        conv(p + "1", out_dim, 2 * s, in_dim, 2 * s * in_dim, transpose=True, gain=1.7 * math.sqrt(s))
        j0 = 2
        if noise:
            conv(p + "2.linear", out_dim, 1, out_dim, out_dim, gain=0.3, bias=False)
            j0 = 3
        for j in range(3):
            q = p + f"{j0 + j}.block.layers."
            alpha(q + "0", out_dim)
            if depthwise:
                conv(q + "1", out_dim, 7, 1, 7, gain=1.2)
            else:
                conv(q + "1", out_dim, 7, out_dim, 7 * out_dim, gain=1.2)
            alpha(q + "2", out_dim)
            conv(q + "3", out_dim, 1, out_dim, out_dim, gain=0.5)
    n = nxt + len(decoder_rates)
    alpha(f"{m}{n}", out_dim)
    conv(f"{m}{n + 1}", 1, 7, out_dim, 7 * out_dim, gain=0.3)
    return w


def make_snac_encoder_weights(encoder_dim: int, encoder_rates: List[int], latent_dim: int, vq_strides: List[int], codebook_dim: int,
                              depthwise: bool = True, seed: int = 0, attn: bool = False) -> Dict[str, torch.Tensor]:
    """Random float32 ENCODE-side parameters (``encoder.*`` and every quantizer's ``in_proj``; reference module paths, MLX layouts): merge with
    ``make_snac_weights`` for a whole model."""
    g = torch.Generator().manual_seed(seed + 104729)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin_g, fan_in, gain=1.0):
        v0 = (torch.rand(cout, k, cin_g, generator=g) * 2 - 1) * math.sqrt(1 / fan_in) * gain
        gw = torch.sqrt((v0 ** 2).sum(dim=(1, 2), keepdim=True))
        w[name + ".weight_g"] = gw * (1.0 + 0.1 * torch.randn(gw.shape, generator=g))
        w[name + ".weight_v"] = v0 / (gw + 1e-12)
        w[name + ".bias"] = 0.02 * torch.randn(cout, generator=g)

    def alpha(name, c):
        w[name + ".alpha"] = (1.0 + 0.3 * torch.randn(1, c, 1, generator=g)).abs() + 0.05

    e = "encoder.block.layers."
    conv(e + "0", encoder_dim, 7, 1, 7, gain=2.5)
    d = encoder_dim
    for i, s in enumerate(encoder_rates):
        d *= 2
        p = f"{e}{i + 1}.block.layers."
        for j in range(3):
            q = p + f"{j}.block.layers."
            alpha(q + "0", d // 2)
            if depthwise:
                conv(q + "1", d // 2, 7, 1, 7, gain=1.2)
            else:
                conv(q + "1", d // 2, 7, d // 2, 7 * (d // 2), gain=1.2)
            alpha(q + "2", d // 2)
            conv(q + "3", d // 2, 1, d // 2, d // 2, gain=0.5)
        alpha(p + "3", d // 2)
        conv(p + "4", d, 2 * s, d // 2, 2 * s * (d // 2), gain=1.2)
    nxt = len(encoder_rates) + 1
    if attn:
        a = f"{e}{nxt}."
        w[a + "norm.weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        w[a + "norm.bias"] = 0.05 * torch.randn(d, generator=g)
        w[a + "to_qkv.weight"] = (torch.rand(3 * d, d, generator=g) * 2 - 1) * math.sqrt(3.0 / d)
        w[a + "to_out.weight"] = (torch.rand(d, d, generator=g) * 2 - 1) * math.sqrt(1.5 / d)
        nxt += 1
    if depthwise:
        conv(f"{e}{nxt}", d, 7, 1, 7, gain=1.7)
    else:
        conv(f"{e}{nxt}", d, 7, d, 7 * d, gain=1.7)
    assert d == latent_dim, (d, latent_dim)
    for i in range(len(vq_strides)):
        conv(f"quantizer.quantizers.{i}.in_proj", codebook_dim, 1, latent_dim, latent_dim, gain=1.5)
    return w


class _Snake:
    """alpha and 1 / (alpha + 1e-9) (layers.py:123-126), padded to a multiple of 32 channels (conv_gemm / dwconv prologue operands)."""

    def __init__(self, alpha: torch.Tensor, device):
        a = alpha.reshape(-1).float()
        cp = round_up(a.numel(), 32)
        al, ib = torch.ones(cp), torch.zeros(cp)
        al[: a.numel()] = a
        ib[: a.numel()] = torch.reciprocal(a + 1e-9)
        self.alpha, self.inv = al.to(device), ib.to(device)
        # conv_gemm's plain Snake prologue computes 1 / alpha in the kernel (v_rcp_f32 + one Newton step) and is the fast instantiation; the
        # explicit 1 / (alpha + 1e-9) table selects the extended one (more registers, ~20 % slower).  For |alpha| >= 1e-2 the two coefficients
        # differ by <= 1e-7 relative (one float32 ulp), far inside the parity bar, so the table is only handed over when an alpha is tiny.
        self.inv_conv = self.inv if float(a.abs().min()) < 1e-2 else None


def _wn(w: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    v, g = w[name + ".weight_v"].double(), w[name + ".weight_g"].double()
    return (g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float()


class _Quantizer:
    """``ResidualVectorQuantize`` decode side: ``from_codes`` (vq.py:116-137)."""

    def __init__(self, w: Dict[str, torch.Tensor], vq_strides: List[int], codebook_size: int, device):
        self.vq_strides, self.codebook_size, self.device = list(vq_strides), codebook_size, device
        self.n_codebooks = len(vq_strides)
        tabs = []
        for i in range(self.n_codebooks):
            p = f"quantizer.quantizers.{i}."
            wt = _wn(w, p + "out_proj")[:, 0, :].double()  # [D, d]
            tabs.append((w[p + "codebook.weight"].double() @ wt.t() + w[p + "out_proj.bias"].double()).float())
        self.table = torch.cat(tabs, 0).contiguous().to(device)
        self.offs = torch.tensor([i * codebook_size for i in range(self.n_codebooks)], dtype=torch.int32, device=device)
        self.latent_dim = self.table.shape[1]
        # encode side (present when the checkpoint carries the in_proj convs)
        self.in_proj = None
        if all(f"quantizer.quantizers.{i}.in_proj.weight_v" in w for i in range(self.n_codebooks)):
            self.in_proj, self.search = [], []
            for i, s in enumerate(self.vq_strides):
                p = f"quantizer.quantizers.{i}.in_proj"
                wi = _wn(w, p)                                    # [d, 1, D]
                # avg_pool(stride s) then in_proj = one tap over rows regrouped s at a time: weights W / s, tiled s times (vq.py:25-32)
                self.in_proj.append(ops.pack_conv((wi / s).repeat(1, 1, s).contiguous() if s > 1 else wi, w.get(p + ".bias"), device, f16=True))
                cb = w[f"quantizer.quantizers.{i}.codebook.weight"].float()
                cn = cb / torch.clamp(torch.sqrt((cb * cb).sum(1, keepdim=True)), min=1e-12)      # normalize() of vq.py:140-143
                t = cn[None].contiguous()
                # |c|^2 / 2 of an L2-normalised row is the CONSTANT 0.5: with it the arg-min of the score is exactly the arg-max of the cosine whatever |e| is (the float32
                # rounding spread of a computed |c|^2, ~1e-7, would bias the decision by 1e-7 / |e| for small-norm residual projections: ADVICE r5); an all-zero row keeps 0
                half = torch.where((cb * cb).sum(1) > 1e-24, torch.full((cb.shape[0],), 0.5), ((t * t).sum(-1) / 2)[0])[None].contiguous()
                self.search.append((t.to(device), t.transpose(1, 2).contiguous().to(device), half.to(device)))
            self.codebook_dim = self.search[0][0].shape[2]
            self.neg_table = (-self.table).contiguous()

    def __call__(self, z, return_margins: bool = False, force=None):
        """``ResidualVectorQuantize.__call__`` (vq.py:102-113): z [B, D, T] -> (z_q [B, D, T], codes: list of int64 [B, T / stride_i]).
        ``return_margins`` appends the cosine gap between the best and the second-best codeword of every decision (list of [B, T / stride_i]).
        ``force`` = (masks, codes), lists of [B, T / stride_i] per level: parity-test hook -- where a mask is set the given code replaces the search result
        (the search still runs and reports its margin): re-synchronises the residual chain with another build's at a knife edge."""
        if self.in_proj is None:
            raise ValueError("this SNAC was loaded without quantizer in_proj weights (decode-only checkpoint): the codebook search cannot run")
        z = torch.as_tensor(z, dtype=torch.float32).to(self.device).transpose(1, 2)   # channels-last rows for the kernels
        B, T, D = z.shape
        if D != self.latent_dim:
            raise ValueError(f"quantizer: z must be [B, {self.latent_dim}, T], got {tuple(z.transpose(1, 2).shape)}")
        for s in self.vq_strides:
            if T % s:
                raise ValueError(f"quantizer: {T} frames are not a whole number of stride-{s} groups (SNAC.preprocess pads the audio so that they are)")
        residual = z.contiguous().clone()
        d = self.codebook_dim
        codes, margins = [], []
        for i, s in enumerate(self.vq_strides):
            Ts = T // s
            ze = torch.empty((B, Ts, d), dtype=torch.float32, device=self.device)
            ops.conv_gemm(residual.view(B, Ts, s * D), self.in_proj[i], ze, precision=4)
            rows = ze.view(B * Ts, d)
            c, m = ops.rvq_encode(rows, *self.search[i], margins=True)
            if return_margins:
                margins.append(m.view(B, Ts) / torch.clamp(torch.sqrt((rows * rows).sum(1)).view(B, Ts), min=1e-30))
            c = c.view(B, Ts)
            if force is not None:
                c = torch.where(torch.as_tensor(force[0][i]).to(self.device), torch.as_tensor(force[1][i]).to(self.device, torch.int32), c)
            codes.append(c.to(torch.int64))
            if i + 1 < self.n_codebooks:
                ids = (torch.repeat_interleave(c, s, dim=1) if s > 1 else c).reshape(B, T, 1).contiguous()
                ops.embed_sum(self.neg_table, ids, residual, slot_offset=self.offs[i:i + 1], add=residual)
        z_q = self.from_codes(codes)
        return (z_q, codes, margins) if return_margins else (z_q, codes)

    def from_codes(self, codes: List[torch.Tensor]) -> torch.Tensor:
        """codes[i] int [B, T / stride_i] -> z_q [B, D, T]."""
        if len(codes) != self.n_codebooks:
            raise IndexError(f"from_codes: {len(codes)} code tensors given, the model has {self.n_codebooks} quantizers")
        cs = [torch.as_tensor(c).to(self.device) for c in codes]
        T = cs[0].shape[1] * self.vq_strides[0]
        cols = []
        for c, s in zip(cs, self.vq_strides):
            if c.shape[1] * s != T:
                raise ValueError(f"from_codes: level with stride {s} has {c.shape[1]} frames, expected {T // s}")
            if int(c.min()) < 0 or int(c.max()) >= self.codebook_size:
                raise IndexError("from_codes: code out of range")
            cols.append(torch.repeat_interleave(c.to(torch.int32), s, dim=1) if s > 1 else c.to(torch.int32))
        ids = torch.stack(cols, dim=-1).contiguous()  # [B, T, n]
        z = torch.empty((ids.shape[0], T, self.latent_dim), dtype=torch.float32, device=self.device)
        ops.embed_sum(self.table, ids, z, slot_offset=self.offs)
        return z.transpose(1, 2)


class SNAC:
    def __init__(self, sampling_rate=44100, encoder_dim=64, encoder_rates=[3, 3, 7, 7], latent_dim=None, decoder_dim=1536,
                 decoder_rates=[7, 7, 3, 3], attn_window_size=32, codebook_size=4096, codebook_dim=8, vq_strides=[8, 4, 2, 1], noise=True,
                 depthwise=True, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", seed: int = 0, **kwargs):
        """Same arguments as the reference (snac.py:16-31) plus ``weights`` (reference parameter names; omitted: random, like a freshly
        constructed reference model), ``device``, ``seed``."""
        ops.require_gpu()
        self.sampling_rate, self.encoder_dim, self.encoder_rates = sampling_rate, encoder_dim, list(encoder_rates)
        self.decoder_dim, self.decoder_rates = decoder_dim, list(decoder_rates)
        self.latent_dim = encoder_dim * (2 ** len(encoder_rates)) if latent_dim is None else latent_dim
        self.hop_length = int(np.prod(encoder_rates))
        self.n_codebooks, self.codebook_size, self.codebook_dim = len(vq_strides), codebook_size, codebook_dim
        self.vq_strides, self.attn_window_size, self.noise, self.depthwise = list(vq_strides), attn_window_size, noise, depthwise
        self.device = torch.device(device)
        if weights is None:   # a freshly constructed reference model has both halves (the reference's own test encodes with one: codec/tests/test_snac.py)
            weights = make_snac_weights(self.latent_dim, decoder_dim, self.decoder_rates, self.vq_strides, codebook_size, codebook_dim, noise, depthwise, seed,
                                        attn=attn_window_size is not None)
            if self.latent_dim == encoder_dim * 2 ** len(self.encoder_rates):   # (the encoder ends at that width: a different latent_dim cannot be fed by it)
                weights.update(make_snac_encoder_weights(encoder_dim, self.encoder_rates, self.latent_dim, self.vq_strides, codebook_dim, depthwise, seed,
                                                         attn=attn_window_size is not None and self.latent_dim % 64 == 0))
        self.load_weights(weights)

    # ------------------------------------------------------------------ load
    def load_weights(self, weights: Dict[str, torch.Tensor]):
        dev = self.device
        w = {k: torch.as_tensor(v).detach().float().cpu() for k, v in weights.items() if k.startswith(("decoder.", "quantizer.", "encoder."))}

        def conv(name) -> PackedConv:
            return ops.pack_conv(_wn(w, name), w.get(name + ".bias"), dev, f16=True)

        def dw(name) -> Tuple[torch.Tensor, torch.Tensor]:
            return _wn(w, name)[:, :, 0].contiguous().to(dev), w[name + ".bias"].contiguous().to(dev)  # [C, 7], [C]

        def convT(name, stride) -> PackedConv:
            # stored (in, K, out), normalised over all axes but 0, handed to MLX as weight.swapaxes(0, 2) = (out, K, in) (layers.py:103-117)
            return ops.pack_conv_transpose(_wn(w, name).permute(2, 1, 0).contiguous(), w.get(name + ".bias"), stride, dev, f16=True)

        def mid(name):
            return dw(name) if self.depthwise else conv(name)

        self.quantizer = _Quantizer(w, self.vq_strides, self.codebook_size, dev)
        m = "decoder.model.layers."
        if self.depthwise:
            self.in_dw, self.conv_in, nxt = dw(m + "0"), conv(m + "1"), 2
        else:
            self.in_dw, self.conv_in, nxt = None, conv(m + "0"), 1
        self.attn = None
        if self.attn_window_size is not None:   # LocalMHA (attention.py:5-53) between the input convs and the first DecoderBlock (layers.py:185-186)
            a = f"{m}{nxt}."
            dh = 64
            if self.decoder_dim % dh:
                raise ValueError(f"LocalMHA needs decoder_dim ({self.decoder_dim}) to be a multiple of its head size 64")
            inv = 1.0 / (10000 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
            ang = torch.arange(self.attn_window_size, dtype=torch.float32)[:, None] * inv[None, :]
            self.attn = dict(nw=w[a + "norm.weight"].to(dev), nb=w[a + "norm.bias"].to(dev), dh=dh, heads=self.decoder_dim // dh,
                             qkv=ops.pack_conv(w[a + "to_qkv.weight"][:, None, :], None, dev, f16=True),
                             out=ops.pack_conv(w[a + "to_out.weight"][:, None, :], None, dev, f16=True),
                             cos=torch.cos(ang).contiguous().to(dev), sin=torch.sin(ang).contiguous().to(dev))
            nxt += 1
        self.blocks = []
        for i, s in enumerate(self.decoder_rates):
            p = f"{m}{nxt + i}.block.layers."
            j0 = 3 if self.noise else 2
            units = []
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j0 + j}.block.layers."
                units.append(dict(dil=d, s1=_Snake(w[q + "0.alpha"], dev), c1=mid(q + "1"), s2=_Snake(w[q + "2.alpha"], dev), c2=conv(q + "3")))
            self.blocks.append(dict(stride=s, snake=_Snake(w[p + "0.alpha"], dev), up=convT(p + "1", s), cout=self.decoder_dim // 2 ** (i + 1),
                                    noise=conv(p + "2.linear") if self.noise else None, units=units))
        n = nxt + len(self.decoder_rates)
        self.out_snake = _Snake(w[f"{m}{n}.alpha"], dev)
        self.conv_out = conv(f"{m}{n + 1}")
        self.enc = None
        if "encoder.block.layers.0.weight_v" in w:   # the encode half (layers.py:132-156)
            e = "encoder.block.layers."
            w0 = _wn(w, e + "0")                      # [d, 7, 1] -> one tap of 7 "channels" (flattened conv)
            stem = ops.pack_conv(w0.reshape(w0.shape[0], 1, w0.shape[1]).contiguous(), w.get(e + "0.bias"), dev, f16=True)
            blocks = []
            for i, s in enumerate(self.encoder_rates):
                p = f"{e}{i + 1}.block.layers."
                units = []
                for j, d in enumerate((1, 3, 9)):
                    q = p + f"{j}.block.layers."
                    units.append(dict(dil=d, s1=_Snake(w[q + "0.alpha"], dev), c1=mid(q + "1"), s2=_Snake(w[q + "2.alpha"], dev), c2=conv(q + "3")))
                wd = _wn(w, p + "4")                  # [cout, 2 s, cin]: tap j s + r -> (tap j, channel r cin + c)
                cout, k, cin = wd.shape
                if k != 2 * s:
                    raise ValueError(f"encoder block {i}: {k} taps for stride {s} (the reference builds kernel_size = 2 * stride)")
                blocks.append(dict(stride=s, cin=cin, units=units, snake=_Snake(w[p + "3.alpha"].reshape(-1).repeat(s), dev),
                                   down=ops.pack_conv(wd.reshape(cout, 2, s * cin).contiguous(), w.get(p + "4.bias"), dev, f16=True)))
            nxt = len(self.encoder_rates) + 1
            enc_attn = None
            if self.attn_window_size is not None and f"{e}{nxt}.to_qkv.weight" in w:
                a = f"{e}{nxt}."
                dh, dm = 64, self.latent_dim
                if dm % dh:
                    raise ValueError(f"LocalMHA needs the encoder width ({dm}) to be a multiple of its head size 64")
                inv = 1.0 / (10000 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
                ang = torch.arange(self.attn_window_size, dtype=torch.float32)[:, None] * inv[None, :]
                enc_attn = dict(nw=w[a + "norm.weight"].to(dev), nb=w[a + "norm.bias"].to(dev), dh=dh, heads=dm // dh,
                                qkv=ops.pack_conv(w[a + "to_qkv.weight"][:, None, :], None, dev, f16=True),
                                out=ops.pack_conv(w[a + "to_out.weight"][:, None, :], None, dev, f16=True),
                                cos=torch.cos(ang).contiguous().to(dev), sin=torch.sin(ang).contiguous().to(dev))
                nxt += 1
            self.enc = dict(stem=stem, k0=w0.shape[1], dim=w0.shape[0], blocks=blocks, attn=enc_attn, out=mid(f"{e}{nxt}"))
        return self

    @classmethod
    def from_config(cls, config_path, **kwargs) -> "SNAC":
        """snac.py:177-182: a model (random parameters, like the reference's) from a ``config.json``."""
        import json

        with open(config_path, "r") as f:
            config = json.load(f)
        return cls(**config, **kwargs)

    @classmethod
    def from_pretrained(cls, repo_id, device="cuda:0", **kwargs) -> "SNAC":
        """snac.py:184-201 for a LOCAL directory (``config.json`` + ``model.safetensors``); the reference's ``fetch_from_hub`` needs the network."""
        from pathlib import Path

        from safetensors.torch import load_file

        path = Path(repo_id)
        if not path.exists():
            raise FileNotFoundError(f"{repo_id}: SNAC.from_pretrained needs a local directory (no hub access in this build)")
        return cls.from_config(path / "config.json", weights=load_file(str(path / "model.safetensors")), device=device, **kwargs)

    # ------------------------------------------------------------------ reference surface
    def preprocess(self, audio_data):
        """snac.py:67-86: right-pad to a multiple of hop_length * lcm(vq_strides [, attn_window_size])."""
        audio_data = torch.as_tensor(audio_data)
        length = audio_data.shape[-1]
        lcm_value = self.vq_strides[0]
        for s in self.vq_strides[1:]:
            lcm_value = abs(lcm_value * s) // math.gcd(lcm_value, s)
        if self.attn_window_size:   # the windows of LocalMHA must tile the latent frames (snac.py:76-78)
            lcm_value = abs(lcm_value * self.attn_window_size) // math.gcd(lcm_value, self.attn_window_size)
        pad_to = self.hop_length * lcm_value
        right_pad = math.ceil(length / pad_to) * pad_to - length
        return torch.nn.functional.pad(audio_data, (0, right_pad))

    def encoder(self, audio_data, return_stages: bool = False):
        """``Encoder.__call__`` on ``audio_data.moveaxis(1, 2)`` (layers.py:132-156): audio [B, 1, S] -> z [B, latent_dim, T]."""
        if self.enc is None:
            raise ValueError("this SNAC was loaded without encoder weights (decode-only checkpoint)")
        e = self.enc
        x0 = torch.as_tensor(audio_data, dtype=torch.float32).to(self.device)
        if x0.dim() != 3 or x0.shape[1] != 1:
            raise ValueError(f"encoder: audio_data must be [B, 1, samples], got {tuple(x0.shape)}")
        x0 = x0.reshape(x0.shape[0], -1).contiguous()
        B, L = x0.shape
        st = {}

        def staged(L, C, s):
            """Zeroed buffer whose rows [p, p + L) hold a block's activation; regrouped s rows at a time it is the input of the block's strided conv."""
            p = math.ceil(s / 2)
            if L + 2 * p < 2 * s:
                raise ValueError(f"encoder: {L} rows are fewer than one frame of the stride-{s} conv")
            Lout = (L + 2 * p - 2 * s) // s + 1
            rows = round_up(max((Lout + 1) * s, p + L), s)
            buf = torch.zeros((B, rows, C), dtype=torch.float32, device=self.device)
            return buf, buf[:, p:p + L], Lout

        blocks = e["blocks"]
        buf, y, Lout = staged(L, e["dim"], blocks[0]["stride"])
        ops.conv_gemm(x0[:, :, None], e["stem"], y, lout=L, flat=dict(ldx=1, x_off=-(e["k0"] // 2), channels=1), precision=4)
        for bi, blk in enumerate(blocks):
            s, C = blk["stride"], blk["cin"]
            tmp = self._f(B, L, C)
            for u in blk["units"]:
                self._mid(y, u["s1"], u["c1"], tmp, u["dil"])
                self._conv(tmp, u["s2"], u["c2"], y, res=y)
            if return_stages:
                st[f"units{bi}"] = y.clone()
            cout = blk["down"].cout
            if bi + 1 < len(blocks):
                nbuf, ny, nLout = staged(Lout, cout, blocks[bi + 1]["stride"])
            else:
                nbuf, ny, nLout = None, self._f(B, Lout, cout), 0
            sn = blk["snake"]
            ops.conv_gemm(buf.view(B, buf.shape[1] // s, s * C), blk["down"], ny, pad=0, lout=Lout, pre_act=ACT_SNAKE, pre_alpha=sn.alpha,
                          pre_inv_beta=sn.inv_conv, precision=4)
            buf, y, L, Lout = nbuf, ny, Lout, nLout
            if return_stages:
                st[f"block{bi}"] = y.clone()
        if e["attn"] is not None:
            y = self._local_mha(y, e["attn"])
            st["attn"] = y
        z = self._f(B, L, self.latent_dim)
        if self.depthwise:
            ops.dwconv(y, e["out"][0], e["out"][1], z, pad=3)
        else:
            self._conv(y, None, e["out"], z)
        st["latent"] = z
        return (z.transpose(1, 2), st) if return_stages else z.transpose(1, 2)

    def encode(self, audio_data, return_margins: bool = False, force=None):
        """snac.py:96-102: audio [B, 1, S] -> codes (list of int64 [B, T / vq_strides[i]]); the audio is right-padded first (``preprocess``)."""
        out = self.quantizer(self.encoder(self.preprocess(audio_data)), return_margins=return_margins, force=force)
        return (out[1], out[2]) if return_margins else out[1]

    def __call__(self, audio_data, noises: Optional[List[torch.Tensor]] = None):
        """snac.py:88-94: (audio_hat, codes).  The reference slices the LAST axis of the channels-last decoder output (``audio_hat[..., :length]``:
        one channel, so a no-op for length >= 1) -- mirrored as is, like ``decode_stream``."""
        audio_data = torch.as_tensor(audio_data, dtype=torch.float32)
        length = audio_data.shape[-1]
        z_q, codes = self.quantizer(self.encoder(self.preprocess(audio_data)))
        audio_hat = self.decode_latents(z_q, noises=noises)
        return audio_hat[..., :length], codes

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _conv(self, x, sn: Optional[_Snake], pc: PackedConv, y, *, dil=1, res=None, post_act=ACT_NONE):
        kw = dict(dil=dil, pad=(pc.k - 1) * dil // 2, res=res, post_act=post_act, precision=4)
        if sn is not None:
            kw.update(pre_act=ACT_SNAKE, pre_alpha=sn.alpha, pre_inv_beta=sn.inv_conv)
        return ops.conv_gemm(x, pc, y, **kw)

    def _mid(self, x, sn: _Snake, c, y, dil: int):
        """Snake + the k7 conv of a ResidualUnit: depthwise kernel or dense implicit GEMM."""
        if self.depthwise:
            wk, b = c
            return ops.dwconv(x, wk, b, y, pad=3 * dil, dil=dil, pre_alpha=sn.alpha, pre_inv=sn.inv)
        return self._conv(x, sn, c, y, dil=dil)

    def decode_latents(self, z, noises: Optional[List[torch.Tensor]] = None, return_stages: bool = False):
        """z [B, latent_dim, T] -> audio [B, T', 1] (``self.decoder(z.moveaxis(1, 2))``, snac.py:106).  ``noises[i]`` [B, 1, channels_i] replaces
        DecoderBlock i's ``mx.random.normal((B, 1, T))`` -- whose "T" is the channel count of the channels-last activation (module docstring);
        omitted: drawn on the device."""
        z = torch.as_tensor(z, dtype=torch.float32).to(self.device)
        x = z.transpose(1, 2).contiguous()
        B, T, _ = x.shape
        st = {}
        if self.in_dw is not None:
            t0 = self._f(B, T, self.latent_dim)
            ops.dwconv(x, self.in_dw[0], self.in_dw[1], t0, pad=3)
            x = t0
        h = self._f(B, T, self.decoder_dim)
        self._conv(x, None, self.conv_in, h)
        st["conv_in"] = h
        if self.attn is not None:
            h = self._local_mha(h, self.attn)
            st["attn"] = h
        for bi, blk in enumerate(self.blocks):
            s, cout, taps = blk["stride"], blk["cout"], blk["up"].k
            p = math.ceil(s / 2)
            Lin = h.shape[1]
            Lout = (Lin - 1) * s - 2 * p + 2 * s + 1   # + 1: the reference's groups-as-output_padding slip (module docstring)
            y = self._f(B, Lout, cout)
            ops.conv_gemm(h, blk["up"], y, pad=taps - 1, lout=Lin + taps - 1, up=dict(s=s, p=p, cout=cout, lout=Lout), pre_act=ACT_SNAKE,
                          pre_alpha=blk["snake"].alpha, pre_inv_beta=blk["snake"].inv_conv, precision=4)
            tmp = torch.empty_like(y)
            if blk["noise"] is not None:
                if noises is not None:
                    nz = torch.as_tensor(noises[bi], dtype=torch.float32).to(self.device)
                    assert nz.shape == (B, 1, cout), (nz.shape, (B, 1, cout))
                else:
                    nz = torch.randn((B, 1, cout), dtype=torch.float32, device=self.device)
                self._conv(y, None, blk["noise"], tmp)
                y.addcmul_(nz, tmp)                    # x + noise * linear(x): one elementwise pass over [B, Lout, cout]
            for u in blk["units"]:
                self._mid(y, u["s1"], u["c1"], tmp, u["dil"])
                self._conv(tmp, u["s2"], u["c2"], y, res=y)
            h = y
            st[f"block{bi}"] = h
        out = self._f(B, h.shape[1], 1)
        self._conv(h, self.out_snake, self.conv_out, out, post_act=ACT_TANH)
        return (out, st) if return_stages else out

    def _local_mha(self, h: torch.Tensor, a: dict) -> torch.Tensor:
        """``LocalMHA.__call__`` (attention.py:19-53) on channels-last [B, T, C]: LayerNorm -> to_qkv -> heads of 64 channels, attention INSIDE
        windows of ``attn_window_size`` positions (rotate-half rotary embedding with the position inside the window, no mask) -> to_out + x.  The
        windows are the batch items of one attention launch (a window's rows are contiguous: [B, T, 3C] viewed as [B * windows, window, 3C])."""
        ws = self.attn_window_size
        B, T, C = h.shape
        if T % ws:
            raise ValueError(f"LocalMHA: {T} positions are not a whole number of windows of {ws} (the reference's reshape fails the same way)")
        H, dh = a["heads"], a["dh"]
        xn = self._f(B, T, C)
        ops.layernorm(h, xn, weight=a["nw"], bias=a["nb"], eps=1e-5)
        qkv = self._f(B, T, 3 * C)
        ops.conv_gemm(xn, a["qkv"], qkv, precision=4, flatten=True)
        win = qkv.view(B * (T // ws), ws, 3 * C)
        q, k, v = win[:, :, :C], win[:, :, C:2 * C], win[:, :, 2 * C:]
        ops.head_norm_rope(q, q, heads=H, dh=dh, cos=a["cos"], sin=a["sin"], pos0=0, interleaved=False, second=(k, k, H, None))
        att = self._f(B * (T // ws), ws, C)
        ops.flash_attention(q, k, v, att, heads=H, kv_heads=H, dh=dh, causal=False, mode=1)
        out = self._f(B, T, C)
        ops.conv_gemm(att.view(B, T, C), a["out"], out, res=h, precision=4, flatten=True)
        return out

    def decode(self, codes: List[torch.Tensor], noises: Optional[List[torch.Tensor]] = None):
        """codes[i] int [B, T / vq_strides[i]] -> audio [B, T', 1] (snac.py:104-107)."""
        return self.decode_latents(self.quantizer.from_codes(codes), noises=noises)

    def decode_stream(self, codes: List[torch.Tensor], prev_codes: Optional[List[torch.Tensor]] = None, context_frames: int = 8):
        """snac.py:109-165: decode with ``context_frames`` of previous codes in front, return only the new samples + the next context."""
        codes = [torch.as_tensor(c) for c in codes]
        new_context = [c[:, -context_frames:] if c.shape[1] > context_frames else c for c in codes]
        if prev_codes is None:
            return self.decode(codes), new_context
        combined = []
        for i, (prev, new) in enumerate(zip(prev_codes, codes)):
            layer_context = max(1, context_frames // self.vq_strides[i])
            prev = torch.as_tensor(prev)
            if prev.shape[1] > layer_context:
                prev = prev[:, -layer_context:]
            combined.append(torch.cat([prev.to(new.device), new], dim=1))
        full_audio = self.decode(combined)
        context_samples = context_frames * self.hop_length
        # the reference slices the LAST axis of the [B, T', 1] decoder output (``full_audio[..., context_samples:]``): with one channel that
        # axis has length 1, so for context_samples >= 1 it returns the whole signal.  Mirrored as is.
        new_audio = full_audio[..., context_samples:] if full_audio.shape[-1] > context_samples else full_audio
        return new_audio, new_context
