"""Descript Audio Codec (waveform -> codes / latents -> waveform) on MI355X: host schedule over the HIP kernels.

Mirrors ``mlx_audio/codec/models/descript/dac.py`` + ``nn/layers.py`` + ``nn/quantize.py`` (constructor arguments, ``preprocess``, ``encode``, ``decode``,
``__call__``, ``quantizer(z, n_quantizers)``, ``quantizer.from_codes``), with the reference's op-by-op graph collapsed into:
  * ``ResidualVectorQuantize.from_codes`` (quantize.py:130-139): the per-codebook ``out_proj(codebook[code])`` is a table lookup --
    ``codebook @ W^T + b`` is folded once at load into one ``[n_codebooks * codebook_size, latent_dim]`` table and a frame is ONE ``embed_sum``
    launch (sum of n_codebooks table rows, in codebook order like the reference's running sum);
  * every ``Snake1d`` (layers.py:123-136) is the PROLOGUE of the conv that consumes it (alpha and 1 / (alpha + 1e-9) precomputed at load);
  * ``WNConv1d`` / ``WNConvTranspose1d`` (layers.py:17-120): weight norm folded at load; convs are implicit GEMMs; the transposed convs
    (K = 2 stride) run polyphase as 2-tap stride-1 GEMMs with a strided store; residual adds and the final ``tanh`` are epilogues.
Reference quirk preserved: ``WNConvTranspose1d`` hands ``groups = 1`` to MLX's ``output_padding`` slot (positional order), so each transposed
conv emits one extra sample: T frames -> lengths pinned by the reference's tests (250 -> 80 043, 375 -> 120 043, 430 -> 220 235).


Encode side (round 5; dac.py:36-81, 184-192, nn/quantize.py:17-127), from the same kernels:
  * ``Encoder``: the 1 -> d_model k7 conv runs FLATTENED over the contiguous samples (7 taps = 7 "channels" of a one-tap conv, zero outside the
    signal); a ``ResidualUnit`` is two launches (Snake prologue, residual epilogue), as in the decoder; the strided ``WNConv1d(K = 2 s, stride s,
    padding ceil(s / 2))`` that ends an ``EncoderBlock`` is a TWO-tap conv over rows regrouped ``[rows / s, s * C]`` -- a free view of a zeroed
    buffer that holds the block's activation behind ``padding`` zero rows (Snake(0) = 0 keeps the padding zero through the prologue);
  * ``ResidualVectorQuantize.__call__``: per codebook ``in_proj`` (1x1 conv to 8 dims) -> nearest L2-normalised codeword (``mi355_rvq_encode`` over the
    normalised codebook: arg-max of the cosine; the reference's ``|e|^2 - 2 e.c + |c|^2`` on normalised vectors has the same arg-min) -> the
    residual update ``residual -= out_proj(codebook[idx])`` as one ``embed_sum`` over the NEGATED folded table; ``z_q`` is ``from_codes`` of the result.
Weights: float32 checkpoints are held as fp16 MFMA images, activations split fp16 hi + lo (``precision = 4``); deviation from the float32 oracle
asserted in ``tests/test_dac_gpu.py`` / ``tests/test_codec_encode_gpu.py``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from .... import ops
from ....ops import ACT_NONE, ACT_SNAKE, ACT_TANH, PackedConv, round_up
from .base import CodecMixin, DACFile  # noqa: F401


def make_dac_weights(decoder_dim: int, decoder_rates: List[int], latent_dim: int, n_codebooks: int, codebook_size: int, codebook_dim: int,
                     seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 decode-side parameters of the shapes ``DAC(...)`` allocates (reference module paths, MLX layouts)."""
    g = torch.Generator().manual_seed(seed)

    def conv(name, cout, k, cin, transpose=False, gain=1.0):
        scale = math.sqrt(1 / (cin * k))
        v0 = (torch.rand(cout, k, cin, generator=g) * 2 - 1) * scale * gain
        dims = (0, 1) if transpose else (1, 2)
        gw = torch.sqrt((v0 ** 2).sum(dim=dims, keepdim=True))
        w[name + ".weight_g"] = gw * (1.0 + 0.1 * torch.randn(gw.shape, generator=g))
        w[name + ".weight_v"] = v0 / (gw + 1e-12)
        w[name + ".bias"] = 0.02 * torch.randn(cout, generator=g)

    def alpha(name, c):
        w[name + ".alpha"] = (1.0 + 0.3 * torch.randn(1, 1, c, generator=g)).abs() + 0.05

    w: Dict[str, torch.Tensor] = {}
    for i in range(n_codebooks):
        p = f"quantizer.quantizers.{i}."
        w[p + "codebook.weight"] = torch.randn(codebook_size, codebook_dim, generator=g)
        conv(p + "out_proj", latent_dim, 1, codebook_dim, gain=1.0 / math.sqrt(n_codebooks))
    conv("decoder.model.layers.0", decoder_dim, 7, latent_dim, gain=1.7)
    out_dim = decoder_dim
    for i, s in enumerate(decoder_rates):
        in_dim, out_dim = decoder_dim // 2 ** i, decoder_dim // 2 ** (i + 1)
        p = f"decoder.model.layers.{i + 1}.block.layers."
        alpha(p + "0", in_dim)
        conv(p + "1", out_dim, 2 * s, in_dim, transpose=True, gain=1.7 * math.sqrt(s))
        for j in range(3):
            q = p + f"{j + 2}.block.layers."
            alpha(q + "0", out_dim)
            conv(q + "1", out_dim, 7, out_dim, gain=1.2)
            alpha(q + "2", out_dim)
            conv(q + "3", out_dim, 1, out_dim, gain=0.5)
    n = len(decoder_rates)
    alpha(f"decoder.model.layers.{n + 1}", out_dim)
    conv(f"decoder.model.layers.{n + 2}", 1, 7, out_dim, gain=0.3)
    return w


def make_dac_encoder_weights(encoder_dim: int, encoder_rates: List[int], latent_dim: int, n_codebooks: int, codebook_dim: int,
                             seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 ENCODE-side parameters (``encoder.*`` and every quantizer's ``in_proj``; reference module paths, MLX layouts): merge with
    ``make_dac_weights`` for a whole model."""
    g = torch.Generator().manual_seed(seed + 7919)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin, gain=1.0):
        scale = math.sqrt(1 / (cin * k))
        v0 = (torch.rand(cout, k, cin, generator=g) * 2 - 1) * scale * gain
        gw = torch.sqrt((v0 ** 2).sum(dim=(1, 2), keepdim=True))
        w[name + ".weight_g"] = gw * (1.0 + 0.1 * torch.randn(gw.shape, generator=g))
        w[name + ".weight_v"] = v0 / (gw + 1e-12)
        w[name + ".bias"] = 0.02 * torch.randn(cout, generator=g)

    def alpha(name, c):
        w[name + ".alpha"] = (1.0 + 0.3 * torch.randn(1, 1, c, generator=g)).abs() + 0.05

    e = "encoder.block.layers."
    conv(e + "0", encoder_dim, 7, 1, gain=2.5)
    d = encoder_dim
    for i, s in enumerate(encoder_rates):
        d *= 2
        p = f"{e}{i + 1}.block.layers."
        for j in range(3):
            q = p + f"{j}.block.layers."
            alpha(q + "0", d // 2)
            conv(q + "1", d // 2, 7, d // 2, gain=1.2)
            alpha(q + "2", d // 2)
            conv(q + "3", d // 2, 1, d // 2, gain=0.5)
        alpha(p + "3", d // 2)
        conv(p + "4", d, 2 * s, d // 2, gain=1.2)
    n = len(encoder_rates)
    alpha(f"{e}{n + 1}", d)
    conv(f"{e}{n + 2}", latent_dim, 3, d, gain=1.5)
    for i in range(n_codebooks):
        conv(f"quantizer.quantizers.{i}.in_proj", codebook_dim, 1, latent_dim, gain=1.5)
    return w


class _Snake:
    """alpha and 1 / (alpha + 1e-9) (layers.py:123-126), padded to a multiple of 32 channels (conv_gemm prologue operands)."""

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


class _Quantizer:
    """``ResidualVectorQuantize`` decode side: ``from_codes`` (quantize.py:130-139)."""

    def __init__(self, w: Dict[str, torch.Tensor], n_codebooks: int, codebook_size: int, device):
        self.n_codebooks, self.codebook_size, self.device = n_codebooks, codebook_size, device
        tabs, self.codebooks = [], []
        for i in range(n_codebooks):
            p = f"quantizer.quantizers.{i}."
            v, gw = w[p + "out_proj.weight_v"].double(), w[p + "out_proj.weight_g"].double()
            wt = (gw * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True)))[:, 0, :]  # [D, d]
            cb = w[p + "codebook.weight"]
            tabs.append((cb.double() @ wt.t() + w[p + "out_proj.bias"].double()).float())
            self.codebooks.append(cb.float().to(device))
        self.table = torch.cat(tabs, 0).contiguous().to(device)
        self.offs = torch.tensor([i * codebook_size for i in range(n_codebooks)], dtype=torch.int32, device=device)
        self.latent_dim = self.table.shape[1]
        self.codebook_dim = self.codebooks[0].shape[1]
        # encode side (present when the checkpoint carries the in_proj convs): the 1x1 projections, the L2-normalised codebooks in the layouts
        # mi355_rvq_encode reads, and the NEGATED out_proj table for the residual update
        self.in_proj = None
        if all(f"quantizer.quantizers.{i}.in_proj.weight_v" in w for i in range(n_codebooks)):
            self.in_proj, self.search = [], []
            for i in range(n_codebooks):
                p = f"quantizer.quantizers.{i}.in_proj"
                v, gw = w[p + ".weight_v"].double(), w[p + ".weight_g"].double()
                self.in_proj.append(ops.pack_conv((gw * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float(), w.get(p + ".bias"), device, f16=True))
                cb = w[f"quantizer.quantizers.{i}.codebook.weight"].float()
                cn = cb / torch.clamp(torch.sqrt((cb * cb).sum(1, keepdim=True)), min=1e-12)      # normalize() of quantize.py:10-12
                t = cn[None].contiguous()
                # |c|^2 / 2 of an L2-normalised row is the CONSTANT 0.5: with it the arg-min of the score is exactly the arg-max of the cosine whatever |e| is (the float32
                # rounding spread of a computed |c|^2, ~1e-7, would bias the decision by 1e-7 / |e| for small-norm residual projections: ADVICE r5); an all-zero row keeps 0
                half = torch.where((cb * cb).sum(1) > 1e-24, torch.full((cb.shape[0],), 0.5), ((t * t).sum(-1) / 2)[0])[None].contiguous()
                self.search.append((t.to(device), t.transpose(1, 2).contiguous().to(device), half.to(device)))
            self.neg_table = (-self.table).contiguous()

    def __call__(self, z, n_quantizers: Optional[int] = None, return_margins: bool = False, force=None):
        """``ResidualVectorQuantize.__call__`` (quantize.py:90-127): z [B, D, T] ->
        (z_q [B, D, T], codes [B, n, T] int64, latents [B, n * d, T], commitment_loss, codebook_loss).  ``return_margins`` appends the cosine gap
        between the best and the second-best codeword of every decision ([B, n, T]: a gap at float32 rounding level is a knife edge).
        ``force`` = (mask bool [B, n, T], codes int [B, n, T]): parity-test hook -- where the mask is set the given code replaces the search result
        (the search still runs, its margin is still reported), so that a residual chain can be re-synchronised with another build's at a knife edge."""
        if self.in_proj is None:
            raise ValueError("this DAC was loaded without quantizer in_proj weights (decode-only checkpoint): the codebook search cannot run")
        z = torch.as_tensor(z, dtype=torch.float32).to(self.device).transpose(1, 2)   # channels-last rows for the kernels
        B, T, D = z.shape
        if D != self.latent_dim:
            raise ValueError(f"quantizer: z must be [B, {self.latent_dim}, T], got {tuple(z.transpose(1, 2).shape)}")
        n = self.n_codebooks if n_quantizers is None else min(int(n_quantizers), self.n_codebooks)
        if n < 1:
            raise ValueError("quantizer: n_quantizers must be at least 1")
        d = self.codebook_dim
        residual = z.contiguous().clone()
        lat = torch.empty((B, T, n * d), dtype=torch.float32, device=self.device)
        ids, margins = [], []
        for i in range(n):
            ze = lat[:, :, i * d:(i + 1) * d]
            ops.conv_gemm(residual, self.in_proj[i], ze, precision=4)
            rows = lat.view(B * T, -1)[:, i * d:(i + 1) * d]
            c, m = ops.rvq_encode(rows, *self.search[i], margins=True)
            if return_margins:   # score = |c|^2 / 2 - e . c over the normalised codebook: the gap in cosine units is the score gap over |e|
                margins.append((m.view(B, T) / torch.clamp(torch.sqrt((rows * rows).sum(1)).view(B, T), min=1e-30)))
            if force is not None:
                fm, fc = force
                c = torch.where(torch.as_tensor(fm)[:, i].to(self.device).reshape(B * T, 1), torch.as_tensor(fc)[:, i].to(self.device, torch.int32).reshape(B * T, 1), c)
            ids.append(c.view(B, T, 1))
            if i + 1 < n:
                ops.embed_sum(self.neg_table, ids[-1], residual, slot_offset=self.offs[i:i + 1], add=residual)
        codes = torch.cat(ids, 2).permute(0, 2, 1).contiguous().to(torch.int64)      # [B, n, T]
        z_q, z_p, _ = self.from_codes(codes)
        latents = lat.transpose(1, 2)
        # (z_e - codebook[idx])^2 averaged over (d, T) per item and codebook, summed over the codebooks and averaged over the batch; the two
        # losses are the same number in a forward pass (quantize.py:29-30)
        diff = lat.view(B, T, n, d) - z_p.transpose(1, 2).reshape(B, T, n, d)
        loss = (diff * diff).mean(dim=(1, 3)).mean(dim=0).sum()
        out = (z_q, codes, latents, loss, loss.clone())
        return out + (torch.stack(margins, 1),) if return_margins else out

    def from_codes(self, codes):
        """codes int [B, n, T] -> (z_q [B, D, T], z_p [B, n * d, T], codes)."""
        codes_d = torch.as_tensor(codes).to(self.device)
        B, n, T = codes_d.shape
        if n > self.n_codebooks:
            raise IndexError(f"from_codes: {n} codebooks given, the model has {self.n_codebooks}")
        if int(codes_d.min()) < 0 or int(codes_d.max()) >= self.codebook_size:
            raise IndexError("from_codes: code out of range")
        ids = codes_d.to(torch.int32).permute(0, 2, 1)  # [B, T, n] view
        z = torch.empty((B, T, self.latent_dim), dtype=torch.float32, device=self.device)
        ops.embed_sum(self.table, ids, z, slot_offset=self.offs[:n])
        z_p = torch.cat([self.codebooks[i][codes_d[:, i, :].long()] for i in range(n)], dim=-1)  # gather: [B, T, n * d]
        return z.transpose(1, 2), z_p.transpose(1, 2), codes


class DAC(CodecMixin):
    def __init__(self, encoder_dim: int = 64, encoder_rates: List[int] = [2, 4, 5, 8], latent_dim: int = None, decoder_dim: int = 1536,
                 decoder_rates: List[int] = [8, 5, 4, 2], n_codebooks: int = 32, codebook_size: int = 1024, codebook_dim: Union[int, list] = 8,
                 sample_rate: int = 44100, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", seed: int = 0, **kwargs):
        """Same arguments as the reference (dac.py:131-178) plus ``weights`` (reference parameter names; omitted: random, like a freshly
        constructed reference model), ``device``, ``seed``."""
        ops.require_gpu()
        if not isinstance(codebook_dim, int):
            if len(set(codebook_dim)) != 1:
                raise NotImplementedError("per-codebook codebook_dim lists with different sizes")
            codebook_dim = codebook_dim[0]
        self.encoder_dim, self.encoder_rates, self.decoder_dim, self.decoder_rates = encoder_dim, list(encoder_rates), decoder_dim, list(decoder_rates)
        self.sample_rate = sample_rate
        self.latent_dim = encoder_dim * (2 ** len(encoder_rates)) if latent_dim is None else latent_dim
        self.hop_length = int(np.prod(encoder_rates))
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        self.device = torch.device(device)
        if weights is None:   # a freshly constructed reference model has both halves (the reference's own tests encode with one: codec/tests/test_descript.py)
            weights = make_dac_weights(decoder_dim, self.decoder_rates, self.latent_dim, n_codebooks, codebook_size, codebook_dim, seed)
            weights.update(make_dac_encoder_weights(encoder_dim, self.encoder_rates, self.latent_dim, n_codebooks, codebook_dim, seed))
        self.load_weights(weights)

    # ------------------------------------------------------------------ load
    def load_weights(self, weights: Dict[str, torch.Tensor]):
        dev = self.device
        w = {k: torch.as_tensor(v).detach().float().cpu() for k, v in weights.items() if k.startswith(("decoder.", "quantizer.", "encoder."))}

        def conv(name) -> PackedConv:
            v, g = w[name + ".weight_v"].double(), w[name + ".weight_g"].double()
            return ops.pack_conv((g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float(), w.get(name + ".bias"), dev, f16=True)

        def convT(name, stride) -> PackedConv:
            v, g = w[name + ".weight_v"].double(), w[name + ".weight_g"].double()
            return ops.pack_conv_transpose((g * v / torch.sqrt((v ** 2).sum(dim=(0, 1), keepdim=True))).float(), w.get(name + ".bias"), stride, dev, f16=True)

        self.quantizer = _Quantizer(w, self.n_codebooks, self.codebook_size, dev)
        self.conv_in = conv("decoder.model.layers.0")
        self.blocks = []
        for i, s in enumerate(self.decoder_rates):
            p = f"decoder.model.layers.{i + 1}.block.layers."
            units = []
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j + 2}.block.layers."
                units.append(dict(dil=d, s1=_Snake(w[q + "0.alpha"], dev), c1=conv(q + "1"), s2=_Snake(w[q + "2.alpha"], dev), c2=conv(q + "3")))
            self.blocks.append(dict(stride=s, snake=_Snake(w[p + "0.alpha"], dev), up=convT(p + "1", s), cout=self.decoder_dim // 2 ** (i + 1), units=units))
        n = len(self.decoder_rates)
        self.out_snake = _Snake(w[f"decoder.model.layers.{n + 1}.alpha"], dev)
        self.conv_out = conv(f"decoder.model.layers.{n + 2}")
        self.enc = None
        if "encoder.block.layers.0.weight_v" in w:   # the encode half (dac.py:36-81)
            e = "encoder.block.layers."
            v, g = w[e + "0.weight_v"].double(), w[e + "0.weight_g"].double()
            w0 = (g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float()   # [d, 7, 1] -> one tap of 7 "channels" (flattened conv)
            stem = ops.pack_conv(w0.reshape(w0.shape[0], 1, w0.shape[1]).contiguous(), w.get(e + "0.bias"), dev, f16=True)
            blocks = []
            for i, s in enumerate(self.encoder_rates):
                p = f"{e}{i + 1}.block.layers."
                units = []
                for j, d in enumerate((1, 3, 9)):
                    q = p + f"{j}.block.layers."
                    units.append(dict(dil=d, s1=_Snake(w[q + "0.alpha"], dev), c1=conv(q + "1"), s2=_Snake(w[q + "2.alpha"], dev), c2=conv(q + "3")))
                v, g = w[p + "4.weight_v"].double(), w[p + "4.weight_g"].double()
                wd = (g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float()   # [cout, 2 s, cin]: tap j s + r -> (tap j, channel r cin + c)
                cout, k, cin = wd.shape
                if k != 2 * s:
                    raise ValueError(f"encoder block {i}: {k} taps for stride {s} (the reference builds kernel_size = 2 * stride)")
                down = ops.pack_conv(wd.reshape(cout, 2, s * cin).contiguous(), w.get(p + "4.bias"), dev, f16=True)
                blocks.append(dict(stride=s, cin=cin, units=units, snake=_Snake(w[p + "3.alpha"].reshape(-1).repeat(s), dev), down=down))
            ne = len(self.encoder_rates)
            self.enc = dict(stem=stem, k0=w0.shape[1], dim=w0.shape[0], blocks=blocks, snake=_Snake(w[f"{e}{ne + 1}.alpha"], dev), out=conv(f"{e}{ne + 2}"))
        return self

    @classmethod
    def from_pretrained(cls, repo_id: str, device="cuda:0") -> "DAC":
        """dac.py:251-270 for a LOCAL directory (``config.json`` = the constructor's keyword arguments, ``model.safetensors``); the reference's
        ``fetch_from_hub`` needs the network, which this build does not have."""
        import json
        from pathlib import Path

        from safetensors.torch import load_file

        path = Path(repo_id)
        if not path.exists():
            raise FileNotFoundError(f"{repo_id}: DAC.from_pretrained needs a local directory (no hub access in this build)")
        with open(path / "config.json") as f:
            config = json.load(f)
        return cls(**config, weights=load_file(str(path / "model.safetensors")), device=device)

    # ------------------------------------------------------------------ reference surface
    def preprocess(self, audio_data, sample_rate):
        if sample_rate is None:
            sample_rate = self.sample_rate
        assert sample_rate == self.sample_rate
        audio_data = torch.as_tensor(audio_data)
        length = audio_data.shape[-1]
        right_pad = math.ceil(length / self.hop_length) * self.hop_length - length
        return torch.nn.functional.pad(audio_data, (0, right_pad))

    def encoder(self, audio_data, return_stages: bool = False):
        """``Encoder.__call__`` on ``audio_data.moveaxis(1, 2)`` (dac.py:57-81, 189): audio [B, 1, S] -> z [B, latent_dim, T]."""
        if self.enc is None:
            raise ValueError("this DAC was loaded without encoder weights (decode-only checkpoint)")
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
                self._conv(y, u["s1"], u["c1"], tmp, dil=u["dil"])
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
        z = self._f(B, L, self.latent_dim)
        self._conv(y, e["snake"], e["out"], z)
        st["latent"] = z
        return (z.transpose(1, 2), st) if return_stages else z.transpose(1, 2)

    def encode(self, audio_data, n_quantizers: int = None, return_margins: bool = False, force=None):
        """dac.py:184-192: audio [B, 1, S] -> (z [B, D, T], codes [B, n, T], latents [B, n * d, T], commitment_loss, codebook_loss)."""
        return self.quantizer(self.encoder(audio_data), n_quantizers, return_margins=return_margins, force=force)

    def __call__(self, audio_data, sample_rate: int = None, n_quantizers: int = None, use_rvq: bool = True, return_loss: bool = False):
        """dac.py:207-239 (the reference slices the LAST axis of the channels-last decoder output, ``x[..., :length]`` -- one channel, so a no-op
        for length >= 1; mirrored as is, like ``SNAC.decode_stream``)."""
        audio_data = torch.as_tensor(audio_data, dtype=torch.float32)
        length = audio_data.shape[-1]
        audio_data = self.preprocess(audio_data, sample_rate)
        codes = latents = commitment_loss = codebook_loss = None
        if use_rvq:
            z, codes, latents, commitment_loss, codebook_loss = self.encode(audio_data, n_quantizers)
        else:
            z = self.encoder(audio_data)
        x = self.decode(z)
        if return_loss:
            # mx.losses.mse(x, audio_data) broadcasts [B, T', 1] against [B, 1, S] in the reference; restated on matching layouts
            n = min(x.shape[1], audio_data.shape[-1])
            return ((x[:, :n, 0] - audio_data.to(self.device)[:, 0, :n]) ** 2).mean()
        return {"audio": x[..., :length], "z": z, "codes": codes, "latents": latents, "vq/commitment_loss": commitment_loss, "vq/codebook_loss": codebook_loss}

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _conv(self, x, sn: Optional[_Snake], pc: PackedConv, y, *, dil=1, res=None, post_act=ACT_NONE):
        kw = dict(dil=dil, pad=(pc.k - 1) * dil // 2, res=res, post_act=post_act, precision=4)
        if sn is not None:
            kw.update(pre_act=ACT_SNAKE, pre_alpha=sn.alpha, pre_inv_beta=sn.inv_conv)
        return ops.conv_gemm(x, pc, y, **kw)

    def decode(self, z, return_stages: bool = False):
        """z [B, latent_dim, T] -> audio [B, T', 1] (dac.py:193-194)."""
        z = torch.as_tensor(z, dtype=torch.float32).to(self.device)
        x = z.transpose(1, 2).contiguous()
        B, T, _ = x.shape
        st = {}
        h = self._f(B, T, self.decoder_dim)
        self._conv(x, None, self.conv_in, h)
        st["conv_in"] = h
        for bi, blk in enumerate(self.blocks):
            s, cout, taps = blk["stride"], blk["cout"], blk["up"].k
            p = math.ceil(s / 2)
            Lin = h.shape[1]
            Lout = (Lin - 1) * s - 2 * p + 2 * s + 1   # + 1: the reference's groups-as-output_padding slip (module docstring)
            y = self._f(B, Lout, cout)
            ops.conv_gemm(h, blk["up"], y, pad=taps - 1, lout=Lin + taps - 1, up=dict(s=s, p=p, cout=cout, lout=Lout), pre_act=ACT_SNAKE,
                          pre_alpha=blk["snake"].alpha, pre_inv_beta=blk["snake"].inv_conv, precision=4)
            tmp = torch.empty_like(y)
            for u in blk["units"]:
                self._conv(y, u["s1"], u["c1"], tmp, dil=u["dil"])
                self._conv(tmp, u["s2"], u["c2"], y, res=y)
            h = y
            st[f"block{bi}"] = h
        out = self._f(B, h.shape[1], 1)
        self._conv(h, self.out_snake, self.conv_out, out, post_act=ACT_TANH)
        return (out, st) if return_stages else out
