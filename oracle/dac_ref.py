"""PyTorch-CPU restatement of the Descript Audio Codec, decode and encode paths (TEST ORACLE, not product).

Follows /root/reference/mlx_audio/codec/models/descript statement by statement:
  * ``nn/layers.py:8-14, 17-62``    WNConv1d: weight = g * v / ||v|| (norm over all axes but 0)
  * ``nn/layers.py:65-120``         WNConvTranspose1d: norm over all axes but 2; ``mx.conv_transpose1d(x, weight, stride, padding, dilation,
                                    groups)`` -- MLX's positional order is (..., dilation, output_padding, groups) (cf. bigvgan/conv.py:108-110),
                                    so ``groups = 1`` lands in ``output_padding``: every transposed conv yields ONE MORE output sample than
                                    its padding formula says.  The reference's own tests pin the resulting lengths (codec/tests/
                                    test_descript.py:41-42, 74-75, 107-108: 250 frames -> 80 043, 375 -> 120 043, 430 -> 220 235 samples);
                                    restated as ``output_padding=1``.
  * ``nn/layers.py:123-136``        snake(x, alpha) = x + 1 / (alpha + 1e-9) * sin(alpha x)^2
  * ``dac.py:16-33``                ResidualUnit: snake, conv k7 (dilation d, padding 3 d), snake, conv k1, + x
  * ``dac.py:84-129``               DecoderBlock / Decoder: conv k7 -> blocks (snake, convT K = 2 s, padding ceil(s / 2), three units with
                                    dilations 1 / 3 / 9) -> snake -> conv k7 -> tanh
  * ``nn/quantize.py:42-46, 130-139`` ResidualVectorQuantize.from_codes: codebook lookup, out_proj (1x1 WNConv), sum over codebooks

Encode side (round 5):
  * ``dac.py:16-33, 36-81``         EncoderBlock: three units (dilations 1 / 3 / 9) at dim / 2, snake, WNConv1d(K = 2 s, stride s, padding ceil(s / 2));
                                    Encoder: conv k7 (1 -> d_model), the blocks (d_model doubles per block), snake, conv k3 -> latent
  * ``nn/quantize.py:10-12, 17-62`` VectorQuantize: in_proj (1x1), L2-normalised encodings / codebook, ``dist = |e|^2 - 2 e c^T + |c|^2``,
                                    ``(-dist).argmax(1)`` (first maximum), out_proj of the UN-normalised codeword
  * ``nn/quantize.py:90-127``       ResidualVectorQuantize.__call__: residual loop, codes stacked on axis 1, latents concatenated on axis 1, the two
                                    (numerically equal) losses as per-item means, batch-averaged, summed over the codebooks
pinned the same way: ``DAC.encode`` of the reference's own modules on a seeded checkpoint (``ref_dac_encode.npz``; every code equal, latents 2e-5).

Parameter names are the reference's module paths (``decoder.model.layers.N...``, ``quantizer.quantizers.N.codebook.weight`` ...), layouts MLX's
(conv ``[out, K, in]``).  Arithmetic float32 (float64 on request) on the parameters as given (the published checkpoints are float32).
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files
(imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) and tests/test_reference_fixtures_cpu.py holds
this oracle to the result -- ``DAC.quantizer.from_codes`` + ``DAC.decode`` on a small seeded checkpoint (all four rates): latents 1e-5, waveform 2e-5.  The reference's own tests hold shape / length pins only
(reproduced in tests/test_oracle_golden.py and the GPU tests); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def wn_conv_weight(g: Tensor, v: Tensor) -> Tensor:
    return g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))


def wn_convT_weight(g: Tensor, v: Tensor) -> Tensor:
    return g * v / torch.sqrt((v ** 2).sum(dim=(0, 1), keepdim=True))


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """x [B, T, C], alpha [1, 1, C]."""
    return x + torch.reciprocal(alpha + 1e-9) * torch.sin(alpha * x) ** 2


class DACDecoderRef:
    def __init__(self, weights: Dict[str, Tensor], decoder_rates: List[int], n_codebooks: int, dtype=torch.float32):
        self.w = {k: v.to(dtype) if v.is_floating_point() else v for k, v in weights.items()}
        self.rates = list(decoder_rates)
        self.n_codebooks = n_codebooks
        self.dtype = dtype

    def _conv(self, x: Tensor, name: str, dilation: int = 1, padding: int = 0) -> Tensor:
        w = wn_conv_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # [out, K, in]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), padding=padding, dilation=dilation).transpose(1, 2)

    def _convT(self, x: Tensor, name: str, stride: int) -> Tensor:
        w = wn_convT_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # [out, K, in]
        return F.conv_transpose1d(x.transpose(1, 2), w.permute(2, 0, 1), self.w.get(name + ".bias"), stride=stride, padding=math.ceil(stride / 2),
                                  output_padding=1).transpose(1, 2)

    def from_codes(self, codes: Tensor) -> Tensor:
        """codes int [B, n, T] -> z_q [B, D, T] (quantize.py:130-139)."""
        z = 0.0
        for i in range(codes.shape[1]):
            p = f"quantizer.quantizers.{i}."
            e = self.w[p + "codebook.weight"][codes[:, i, :].long()]  # [B, T, d]
            z = z + self._conv(e, p + "out_proj")
        return z.transpose(1, 2)

    def decode(self, z: Tensor, return_stages: bool = False):
        """z [B, D, T] -> audio [B, T', 1] (dac.py:193-194: ``self.decoder(z.moveaxis(1, 2))``)."""
        x = z.to(self.dtype).transpose(1, 2)
        st = {}
        x = self._conv(x, "decoder.model.layers.0", padding=3)
        st["conv_in"] = x
        for i, s in enumerate(self.rates):
            p = f"decoder.model.layers.{i + 1}.block.layers."
            x = snake(x, self.w[p + "0.alpha"])
            x = self._convT(x, p + "1", s)
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j + 2}.block.layers."
                y = snake(x, self.w[q + "0.alpha"])
                y = self._conv(y, q + "1", dilation=d, padding=3 * d)
                y = snake(y, self.w[q + "2.alpha"])
                y = self._conv(y, q + "3")
                x = x + y
            st[f"block{i}"] = x
        n = len(self.rates)
        x = snake(x, self.w[f"decoder.model.layers.{n + 1}.alpha"])
        x = torch.tanh(self._conv(x, f"decoder.model.layers.{n + 2}", padding=3))
        return (x, st) if return_stages else x


class DACEncoderRef:
    """``DAC.encode`` (dac.py:184-192): ``Encoder`` + ``ResidualVectorQuantize.__call__``."""

    def __init__(self, weights: Dict[str, Tensor], encoder_rates: List[int], n_codebooks: int, dtype=torch.float32):
        self.w = {k: v.to(dtype) if v.is_floating_point() else v for k, v in weights.items()}
        self.rates = list(encoder_rates)
        self.n_codebooks = n_codebooks
        self.dtype = dtype

    def _conv(self, x: Tensor, name: str, dilation: int = 1, padding: int = 0, stride: int = 1) -> Tensor:
        w = wn_conv_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # [out, K, in]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), stride=stride, padding=padding, dilation=dilation).transpose(1, 2)

    def encoder(self, audio: Tensor, return_stages: bool = False):
        """audio [B, 1, S] -> z [B, D, T] (dac.py:57-81 on ``audio_data.moveaxis(1, 2)``)."""
        x = audio.to(self.dtype).transpose(1, 2)   # [B, S, 1]
        st = {}
        e = "encoder.block.layers."
        x = self._conv(x, e + "0", padding=3)
        for i, s in enumerate(self.rates):
            p = f"{e}{i + 1}.block.layers."
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j}.block.layers."
                y = snake(x, self.w[q + "0.alpha"])
                y = self._conv(y, q + "1", dilation=d, padding=3 * d)
                y = snake(y, self.w[q + "2.alpha"])
                y = self._conv(y, q + "3")
                x = x + y
            st[f"units{i}"] = x
            x = snake(x, self.w[p + "3.alpha"])
            x = self._conv(x, p + "4", stride=s, padding=math.ceil(s / 2))
            st[f"block{i}"] = x
        n = len(self.rates)
        x = snake(x, self.w[f"{e}{n + 1}.alpha"])
        x = self._conv(x, f"{e}{n + 2}", padding=1)
        st["latent"] = x
        return (x.transpose(1, 2), st) if return_stages else x.transpose(1, 2)

    def quantize(self, z: Tensor, n_quantizers: int = None, return_margins: bool = False):
        """z [B, D, T] -> (z_q, codes [B, n, T], latents [B, n d, T], commitment_loss, codebook_loss) (+ the top-2 gap of ``-dist`` per decision)."""
        n = self.n_codebooks if n_quantizers is None else min(n_quantizers, self.n_codebooks)
        residual = z.to(self.dtype)
        z_q = 0
        codes, latents, margins = [], [], []
        commit = cbl = 0
        for i in range(n):
            p = f"quantizer.quantizers.{i}."
            z_e = self._conv(residual.transpose(1, 2), p + "in_proj").transpose(1, 2)       # [B, d, T]
            b, d, t = z_e.shape
            enc = z_e.permute(0, 2, 1).reshape(b * t, d)
            cb = self.w[p + "codebook.weight"]
            en = enc / torch.clamp(torch.sqrt((enc.abs() ** 2).sum(1, keepdim=True)), min=1e-12)
            cn = cb / torch.clamp(torch.sqrt((cb.abs() ** 2).sum(1, keepdim=True)), min=1e-12)
            dist = (en ** 2).sum(1, keepdim=True) - 2 * en @ cn.t() + (cn ** 2).sum(1, keepdim=True).t()
            top = torch.topk(-dist, 2, dim=1).values
            idx = (-dist).argmax(1).reshape(b, t)
            margins.append(((top[:, 0] - top[:, 1]) / 2).reshape(b, t))   # -dist = 2 cos - const: half the gap is the cosine gap
            zq_lat = cb[idx].transpose(1, 2)                               # decode_code: [B, d, T]
            commit = commit + ((z_e - zq_lat) ** 2).mean(dim=(1, 2)).mean()
            cbl = cbl + ((zq_lat - z_e) ** 2).mean(dim=(1, 2)).mean()
            z_q_i = self._conv((z_e + (zq_lat - z_e)).transpose(1, 2), p + "out_proj").transpose(1, 2)
            z_q = z_q + z_q_i
            residual = residual - z_q_i
            codes.append(idx)
            latents.append(z_e)
        out = (z_q, torch.stack(codes, 1), torch.cat(latents, 1), commit, cbl)
        return out + (torch.stack(margins, 1),) if return_margins else out

    def encode(self, audio: Tensor, n_quantizers: int = None):
        return self.quantize(self.encoder(audio), n_quantizers)
