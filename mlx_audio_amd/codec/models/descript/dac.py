"""Descript Audio Codec, decode side (codes / latents -> waveform), on MI355X: host schedule over the HIP kernels.

Mirrors ``mlx_audio/codec/models/descript/dac.py`` + ``nn/layers.py`` + ``nn/quantize.py`` (constructor arguments, ``preprocess``, ``decode``,
``quantizer.from_codes``), with the reference's op-by-op graph collapsed into:
  * ``ResidualVectorQuantize.from_codes`` (quantize.py:130-139): the per-codebook ``out_proj(codebook[code])`` is a table lookup --
    ``codebook @ W^T + b`` is folded once at load into one ``[n_codebooks * codebook_size, latent_dim]`` table and a frame is ONE ``embed_sum``
    launch (sum of n_codebooks table rows, in codebook order like the reference's running sum);
  * every ``Snake1d`` (layers.py:123-136) is the PROLOGUE of the conv that consumes it (alpha and 1 / (alpha + 1e-9) precomputed at load);
  * ``WNConv1d`` / ``WNConvTranspose1d`` (layers.py:17-120): weight norm folded at load; convs are implicit GEMMs; the transposed convs
    (K = 2 stride) run polyphase as 2-tap stride-1 GEMMs with a strided store; residual adds and the final ``tanh`` are epilogues.
Reference quirk preserved: ``WNConvTranspose1d`` hands ``groups = 1`` to MLX's ``output_padding`` slot (positional order), so each transposed
conv emits one extra sample: T frames -> lengths pinned by the reference's tests (250 -> 80 043, 375 -> 120 043, 430 -> 220 235).

The encoder / quantiser-search half (``encode``, ``__call__``) is outside the decode hot path and raises.  Weights: float32 checkpoints are held
as fp16 MFMA images, activations split fp16 hi + lo (``precision = 4``); deviation from the float32 oracle asserted in ``tests/test_dac_gpu.py``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from .... import ops
from ....ops import ACT_NONE, ACT_SNAKE, ACT_TANH, PackedConv, round_up


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


class DAC:
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
        if weights is None:
            weights = make_dac_weights(decoder_dim, self.decoder_rates, self.latent_dim, n_codebooks, codebook_size, codebook_dim, seed)
        self.load_weights(weights)

    # ------------------------------------------------------------------ load
    def load_weights(self, weights: Dict[str, torch.Tensor]):
        dev = self.device
        w = {k: torch.as_tensor(v).detach().float().cpu() for k, v in weights.items() if k.startswith(("decoder.", "quantizer."))}

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
        return self

    # ------------------------------------------------------------------ reference surface
    def preprocess(self, audio_data, sample_rate):
        if sample_rate is None:
            sample_rate = self.sample_rate
        assert sample_rate == self.sample_rate
        audio_data = torch.as_tensor(audio_data)
        length = audio_data.shape[-1]
        right_pad = math.ceil(length / self.hop_length) * self.hop_length - length
        return torch.nn.functional.pad(audio_data, (0, right_pad))

    def encode(self, audio_data, n_quantizers: int = None):
        raise NotImplementedError("DAC.encode (encoder + codebook search) is outside the decode hot path of this build (SURVEY section 8(f).2)")

    def __call__(self, audio_data, sample_rate: int = None, n_quantizers: int = None, use_rvq: bool = True, return_loss: bool = False):
        raise NotImplementedError("DAC.__call__ runs the encoder, which this build does not contain; use quantizer.from_codes + decode")

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
