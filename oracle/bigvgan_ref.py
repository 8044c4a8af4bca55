"""PyTorch-CPU restatement of BigVGAN (TEST ORACLE, not product): mel -> waveform.

Follows ``codec/models/bigvgan`` of the reference:
  * ``bigvgan.py:29-125``      BigVGAN: conv_pre (k7) -> per stage: WNConvTranspose1d (K = 2 x rate, padding (K - rate) / 2) -> mean of the AMP blocks ->
                               activation_post -> conv_post (k7) -> tanh (or clip)
  * ``amp.py:10-96``           AMPBlock1 (x + conv2(act2(conv1(act1(x)))) per dilation), AMPBlock2 (x + conv(act(x)))
  * ``resample.py:17-177``     kaiser_sinc_filter1d, UpSample1d (edge pad, depthwise transposed conv x ratio, trimmed), LowPassFilter1d / DownSample1d
                               (edge pad, depthwise conv at stride ratio), Activation1d = down(act(up(x)))
  * ``activation.py:27-51``    SnakeBeta: x + 1 / (beta + 1e-9) * sin^2(alpha x), parameters in log scale when ``snake_logscale``
  * ``conv.py:7-114``          weight norm: g * v / ||v|| over all axes but the output one (conv) / but the input one (transposed conv), no epsilon
Channels-last throughout like the reference.  ``Snake`` (activation.py:5-24) broadcasts its parameter over the TIME axis of a channels-last tensor
(``alpha[None, :, None]``): it cannot run on the shapes BigVGAN produces and no shipped configuration selects it; only ``snakebeta`` is restated.
Parity status: **pinned to the reference's own modules** -- tests/golden/make_reference_fixtures.py runs the reference's BigVGAN files on a seeded
checkpoint over the numpy stand-in for MLX and tests/test_reference_fixtures_cpu.py holds this oracle to the result; the reference's own tests hold shape
pins only (codec/tests/test_bigvgan.py: 800 mel frames -> 800 x prod(rates) samples; reproduced in tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> Tensor:
    """resample.py:17-46 -> [kernel_size] float32."""
    even = kernel_size % 2 == 0
    half = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.from_numpy(np.kaiser(kernel_size, beta=beta)).to(torch.float32)
    time = (torch.arange(-half, half) + 0.5) if even else (torch.arange(kernel_size) - half)
    time = time.to(torch.float32)
    if cutoff == 0:
        return torch.zeros(kernel_size)
    arg = 2 * cutoff * time
    sinc = torch.where(arg == 0, torch.ones_like(arg), torch.sin(math.pi * arg) / math.pi / arg)
    f = 2 * cutoff * window * sinc
    return (f / f.sum()).to(torch.float32)


def upsample2(x: Tensor, filt: Tensor) -> Tensor:
    """UpSample1d(ratio 2, K 12) on [B, L, C] (resample.py:101-136)."""
    K, ratio = filt.numel(), 2
    pad = K // ratio - 1
    pad_left = pad * ratio + (K - ratio) // 2
    pad_right = pad * ratio + (K - ratio + 1) // 2
    C = x.shape[-1]
    xp = F.pad(x.transpose(1, 2), (pad, pad), mode="replicate")
    w = filt.reshape(1, 1, K).expand(C, 1, K)
    y = ratio * F.conv_transpose1d(xp, w, stride=ratio, groups=C)
    return y[:, :, pad_left:-pad_right].transpose(1, 2)


def downsample2(x: Tensor, filt: Tensor) -> Tensor:
    """DownSample1d(ratio 2, K 12) = LowPassFilter1d at stride 2 (resample.py:49-98, 139-154)."""
    K = filt.numel()
    C = x.shape[-1]
    xp = F.pad(x.transpose(1, 2), (K // 2 - 1, K // 2), mode="replicate")
    return F.conv1d(xp, filt.reshape(1, 1, K).expand(C, 1, K), stride=2, groups=C).transpose(1, 2)


class BigVGANRef:
    def __init__(self, weights: Dict[str, Tensor], config: dict, dtype=torch.float32):
        self.cfg = config
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        if config["activation"] != "snakebeta":
            raise NotImplementedError("only snakebeta (see the module docstring)")

    def _conv(self, x: Tensor, name: str, dilation: int = 1) -> Tensor:
        v, g = self.w[name + ".weight_v"], self.w[name + ".weight_g"]
        w = g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))
        k = w.shape[1]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), padding=(k - 1) * dilation // 2, dilation=dilation).transpose(1, 2)

    def _convT(self, x: Tensor, name: str, stride: int) -> Tensor:
        v, g = self.w[name + ".weight_v"], self.w[name + ".weight_g"]
        w = g * v / torch.sqrt((v ** 2).sum(dim=(0, 1), keepdim=True))   # [out, K, in]
        k = w.shape[1]
        return F.conv_transpose1d(x.transpose(1, 2), w.permute(2, 0, 1), self.w.get(name + ".bias"), stride=stride, padding=(k - stride) // 2).transpose(1, 2)

    def _act(self, x: Tensor, name: str) -> Tensor:
        alpha, beta = self.w[name + ".act.alpha"], self.w[name + ".act.beta"]
        if self.cfg["snake_logscale"]:
            alpha, beta = torch.exp(alpha), torch.exp(beta)
        u = upsample2(x, self.w[name + ".upsample.filter"].reshape(-1))
        a = u + (1.0 / (beta + 1e-9)) * torch.sin(u * alpha) ** 2
        return downsample2(a, self.w[name + ".downsample.lowpass.filter"].reshape(-1))

    def __call__(self, mel: Tensor, return_stages: bool = False):
        """mel [B, num_mels, T] -> audio [B, 1, T * prod(rates)]."""
        cfg = self.cfg
        x = self._conv(mel.to(self.dtype).transpose(1, 2), "conv_pre")
        st = {"conv_pre": x}
        nk = len(cfg["resblock_kernel_sizes"])
        for i, u in enumerate(cfg["upsample_rates"]):
            x = self._convT(x, f"ups.{i}.0", u)
            acc = None
            for j, dils in enumerate(cfg["resblock_dilation_sizes"]):
                p = f"resblocks.{i * nk + j}"
                y = x
                if cfg["resblock"] == "1":
                    for q, d in enumerate(dils):
                        t = self._conv(self._act(y, f"{p}.activations.{2 * q}"), f"{p}.convs1.{q}", dilation=d)
                        y = y + self._conv(self._act(t, f"{p}.activations.{2 * q + 1}"), f"{p}.convs2.{q}")
                else:
                    for q, d in enumerate(dils):
                        y = y + self._conv(self._act(y, f"{p}.activations.{q}"), f"{p}.convs.{q}", dilation=d)
                acc = y if acc is None else acc + y
            x = acc / nk
            st[f"stage{i}"] = x
        x = self._conv(self._act(x, "activation_post"), "conv_post")
        x = torch.tanh(x) if cfg.get("use_tanh_at_final", True) else torch.clamp(x, -1.0, 1.0)
        out = x.transpose(1, 2)
        return (out, st) if return_stages else out
