"""PyTorch-CPU restatement of the Descript Audio Codec decode path (TEST ORACLE, not product).

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
