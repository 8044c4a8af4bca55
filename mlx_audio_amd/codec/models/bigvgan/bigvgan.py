"""BigVGAN vocoder (mel -> waveform) on MI355X behind the reference's interface (``codec/models/bigvgan/bigvgan.py:15-149``).

Same surface: ``BigVGANConfig`` fields, ``BigVGAN(config)``, ``__call__(mel [B, num_mels, T]) -> [B, 1, T * prod(upsample_rates)]``, ``sanitize``
(PyTorch layouts -> MLX layouts, ``num_batches_tracked`` dropped).  Schedule over the HIP kernels:

  * ``WNConv1d`` / ``WNConvTranspose1d`` (conv.py:15-114): weight norm folded at load (float32: the published checkpoints are float32), held as fp16 MFMA
    images, activations split fp16 hi + lo (``precision = 4``); transposed convs (K = 2 x rate) run polyphase as 2-tap stride-1 GEMMs with a strided
    store; residual adds, the mean over the AMP blocks of a stage and the final ``tanh`` are conv epilogues;
  * ``Activation1d`` (resample.py:157-177: 2x Kaiser-sinc up-sampling -> SnakeBeta -> 2x low-pass down-sampling) is ONE kernel, ``mi355_aa_activation``:
    the x window and the activated double-rate signal of a 64 x 64 tile live in LDS, so the double-rate tensor never reaches HBM (the reference
    materialises it twice per activation).

``activation = "snake"`` raises: the reference's ``Snake`` broadcasts its parameter over the time axis of the channels-last tensor (activation.py:17),
which only works when T == C; no shipped configuration selects it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Literal, Optional

import torch

from .... import ops
from ....ops import ACT_NONE, ACT_TANH, PackedConv


@dataclass
class BigVGANConfig:
    num_mels: int
    upsample_rates: List[int]
    upsample_kernel_sizes: List[int]
    upsample_initial_channel: int
    resblock: Literal["1", "2"]
    resblock_kernel_sizes: List[int]
    resblock_dilation_sizes: List[List[int]]
    activation: Literal["snakebeta", "snake"]
    snake_logscale: bool
    use_bias_at_final: bool = True
    use_tanh_at_final: bool = True


def _kaiser_sinc_filter(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """resample.py:17-46 (the module buffers ``upsample.filter`` / ``downsample.lowpass.filter``; checkpoints carry them too)."""
    import numpy as np

    half = kernel_size // 2
    A = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.from_numpy(np.kaiser(kernel_size, beta=beta)).to(torch.float32)
    time = ((torch.arange(-half, half) + 0.5) if kernel_size % 2 == 0 else (torch.arange(kernel_size) - half)).to(torch.float32)
    arg = 2 * cutoff * time
    sinc = torch.where(arg == 0, torch.ones_like(arg), torch.sin(math.pi * arg) / math.pi / arg)
    f = 2 * cutoff * window * sinc
    return (f / f.sum()).to(torch.float32)


def make_bigvgan_weights(cfg: BigVGANConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 parameters of the shapes ``BigVGAN(config)`` allocates (reference module paths, MLX layouts)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin, transpose=False, gain=1.0, bias=True):
        v = (torch.rand(cout, k, cin, generator=g) * 2 - 1) * math.sqrt(1 / (cin * k))
        dims = (0, 1) if transpose else (1, 2)
        w[name + ".weight_v"] = v
        w[name + ".weight_g"] = torch.sqrt((v ** 2).sum(dim=dims, keepdim=True)) * gain * (1.0 + 0.1 * torch.rand(1, generator=g))
        if bias:
            w[name + ".bias"] = torch.randn(cout, generator=g) * 0.02

    def act(name, ch):
        w[name + ".act.alpha"] = torch.randn(ch, generator=g) * 0.3
        w[name + ".act.beta"] = torch.randn(ch, generator=g) * 0.3
        f = _kaiser_sinc_filter(0.25, 0.3, 12).reshape(1, 12, 1)
        w[name + ".upsample.filter"] = f.clone()
        w[name + ".downsample.lowpass.filter"] = f.clone()

    c0 = cfg.upsample_initial_channel
    conv("conv_pre", c0, 7, cfg.num_mels, gain=2.0)
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        conv(f"ups.{i}.0", ch, k, cin, transpose=True, gain=math.sqrt(u))
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            p = f"resblocks.{i * nk + j}"
            if cfg.resblock == "1":
                for q in range(len(dils)):
                    conv(f"{p}.convs1.{q}", ch, rk, ch, gain=0.8)
                    conv(f"{p}.convs2.{q}", ch, rk, ch, gain=0.8)
                    act(f"{p}.activations.{2 * q}", ch)
                    act(f"{p}.activations.{2 * q + 1}", ch)
            else:
                for q in range(len(dils)):
                    conv(f"{p}.convs.{q}", ch, rk, ch, gain=0.8)
                    act(f"{p}.activations.{q}", ch)
    ch = c0 // (2 ** len(cfg.upsample_rates))
    act("activation_post", ch)
    conv("conv_post", 1, 7, ch, gain=0.5, bias=cfg.use_bias_at_final)
    return w


@dataclass
class _Act:
    up: torch.Tensor
    down: torch.Tensor
    alpha: torch.Tensor
    inv_beta: torch.Tensor


class BigVGAN:
    def __init__(self, config: BigVGANConfig, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda", seed: int = 0):
        ops.require_gpu()
        if isinstance(config, dict):
            config = BigVGANConfig(**config)
        if config.activation != "snakebeta":
            raise NotImplementedError("BigVGAN activation 'snake': the reference's Snake broadcasts alpha over the time axis (activation.py:17) and only runs "
                                      "when T == C; only 'snakebeta' is supported")
        if any(k % u for u, k in zip(config.upsample_rates, config.upsample_kernel_sizes)):
            raise NotImplementedError("BigVGAN: upsample kernel sizes must be multiples of their rates (polyphase transposed conv)")
        self.config = config
        self.device = torch.device(device)
        self.num_kernels = len(config.resblock_kernel_sizes)
        self.num_upsamples = len(config.upsample_rates)
        self.load_weights(weights if weights is not None else make_bigvgan_weights(config, seed))

    # ------------------------------------------------------------------ checkpoint handling
    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """bigvgan.py:127-149: drop ``num_batches_tracked``; PyTorch Conv1d (out, in, K) -> (out, K, in), filters (1, 1, K) -> (1, K, 1), transposed convs
        (in, out, K) -> (out, K, in) -- each only when the shape differs from the one this model allocates."""
        want = {k: tuple(v.shape) for k, v in make_bigvgan_weights(self.config, 0).items()}
        out = {}
        for key, v in weights.items():
            if "num_batches_tracked" in key:
                continue
            if ("conv" in key or "lowpass.filter" in key or "upsample.filter" in key) and v.dim() == 3 and tuple(v.shape) != want.get(key):
                v = v.permute(0, 2, 1)
            if "ups." in key and v.dim() == 3 and tuple(v.shape) != want.get(key):
                v = v.permute(1, 2, 0)
            out[key] = v.contiguous()
        return out

    def load_weights(self, weights, strict: bool = True):
        w = {k: torch.as_tensor(v).to(torch.float32) for k, v in dict(weights).items()}
        dev = self.device
        cfg = self.config

        def conv(name) -> PackedConv:
            v, g = w[name + ".weight_v"], w[name + ".weight_g"]
            return ops.pack_conv((g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))).float(), w.get(name + ".bias"), dev, f16=True)

        def convT(name, stride) -> PackedConv:
            v, g = w[name + ".weight_v"], w[name + ".weight_g"]
            return ops.pack_conv_transpose((g * v / torch.sqrt((v ** 2).sum(dim=(0, 1), keepdim=True))).float(), w.get(name + ".bias"), stride, dev, f16=True)

        def act(name) -> _Act:
            alpha, beta = w[name + ".act.alpha"], w[name + ".act.beta"]
            if cfg.snake_logscale:
                alpha, beta = torch.exp(alpha), torch.exp(beta)
            fu, fd = w[name + ".upsample.filter"].reshape(-1), w[name + ".downsample.lowpass.filter"].reshape(-1)
            if fu.numel() != 12 or fd.numel() != 12:
                raise NotImplementedError("BigVGAN: anti-aliasing filters must have 12 taps (ratio 2)")
            return _Act(fu.contiguous().to(dev), fd.contiguous().to(dev), alpha.contiguous().to(dev), (1.0 / (beta + 1e-9)).contiguous().to(dev))

        try:
            self.conv_pre = conv("conv_pre")
            self.ups, self.blocks = [], []
            for i, u in enumerate(cfg.upsample_rates):
                self.ups.append(convT(f"ups.{i}.0", u))
                for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
                    p = f"resblocks.{i * self.num_kernels + j}"
                    if cfg.resblock == "1":
                        units = [dict(dil=d, a1=act(f"{p}.activations.{2 * q}"), c1=conv(f"{p}.convs1.{q}"), a2=act(f"{p}.activations.{2 * q + 1}"),
                                      c2=conv(f"{p}.convs2.{q}")) for q, d in enumerate(dils)]
                    else:
                        units = [dict(dil=d, a1=act(f"{p}.activations.{q}"), c1=conv(f"{p}.convs.{q}"), a2=None, c2=None) for q, d in enumerate(dils)]
                    self.blocks.append(units)
            self.act_post = act("activation_post")
            self.conv_post = conv("conv_post")
        except KeyError as e:
            raise ValueError(f"BigVGAN checkpoint is missing parameter {e}") from e
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ forward
    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _conv(self, x, pc: PackedConv, y, *, dil=1, **kw):
        return ops.conv_gemm(x, pc, y, dil=dil, pad=(pc.k - 1) * dil // 2, precision=4, **kw)

    def _act(self, x, a: _Act, y):
        return ops.aa_activation(x, y, a.up, a.down, a.alpha, a.inv_beta)

    @torch.no_grad()
    def __call__(self, x, *args, return_stages: bool = False, **kwargs):
        """mel [B, num_mels, T] -> waveform [B, 1, T * prod(upsample_rates)] (bigvgan.py:99-125)."""
        cfg = self.config
        mel = torch.as_tensor(x, dtype=torch.float32).to(self.device)
        if mel.dim() != 3 or mel.shape[1] != cfg.num_mels:
            raise ValueError(f"BigVGAN expects [batch, {cfg.num_mels}, frames], got {tuple(mel.shape)}")
        h0 = mel.transpose(1, 2).contiguous()
        B, T, _ = h0.shape
        c0 = cfg.upsample_initial_channel
        h = self._f(B, T, c0)
        self._conv(h0, self.conv_pre, h)
        st = {"conv_pre": h}
        nk = self.num_kernels
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            ch = c0 // (2 ** (i + 1))
            Lin, taps = h.shape[1], k // u
            L = Lin * u
            xu = self._f(B, L, ch)
            ops.conv_gemm(h, self.ups[i], xu, pad=taps - 1, lout=Lin + taps - 1, up=dict(s=u, p=(k - u) // 2, cout=ch, lout=L), precision=4)
            acc = self._f(B, L, ch)
            a, t, bufs = self._f(B, L, ch), self._f(B, L, ch), [self._f(B, L, ch), self._f(B, L, ch)]
            for j in range(nk):
                units = self.blocks[i * nk + j]
                src = xu
                for q, un in enumerate(units):
                    last = q == len(units) - 1
                    # the block's last conv lands in the stage accumulator: + previous blocks (accumulate), x 1 / num_kernels after the last one
                    dst = acc if last else bufs[q % 2]
                    ep = dict(res=src, accumulate=last and j > 0, out_scale=(1.0 / nk) if (last and j == nk - 1) else 1.0)
                    self._act(src, un["a1"], a)
                    if un["c2"] is None:  # AMPBlock2: x + conv(act(x))
                        self._conv(a, un["c1"], dst, dil=un["dil"], **ep)
                    else:                 # AMPBlock1: x + conv2(act2(conv1(act1(x))))
                        self._conv(a, un["c1"], t, dil=un["dil"])
                        self._act(t, un["a2"], a)
                        self._conv(a, un["c2"], dst, **ep)
                    src = dst
            h = acc
            st[f"stage{i}"] = h
        a = self._f(*h.shape)
        self._act(h, self.act_post, a)
        out = self._f(B, h.shape[1], 1)
        self._conv(a, self.conv_post, out, post_act=ACT_TANH if cfg.use_tanh_at_final else ACT_NONE)
        if not cfg.use_tanh_at_final:
            out.clamp_(-1.0, 1.0)
        y = out.transpose(1, 2)
        return (y, st) if return_stages else y
