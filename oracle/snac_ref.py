"""PyTorch-CPU restatement of the SNAC decode and encode paths (TEST ORACLE, not product).

Follows /root/reference/mlx_audio/codec/models/snac statement by statement:
  * ``layers.py:9-60``      WNConv1d: weight = g * v / ||v|| (norm over all axes but 0), ``mx.conv1d(x, w, stride, padding, dilation, groups)``
  * ``layers.py:63-120``    WNConvTranspose1d: norm over all axes but 0 of the STORED ``(in, K, out)`` tensor, then
                            ``mx.conv_transpose1d(x, weight.swapaxes(0, 2), stride, padding, dilation, groups)`` -- MLX's positional order is
                            (..., dilation, output_padding, groups), so ``groups = 1`` lands in ``output_padding`` (the constructor's own
                            ``output_padding = stride % 2`` is never used): every transposed conv yields ONE MORE sample than its padding formula
                            says.  The reference's test pins the result: codes of 59 / 118 / 236 frames -> 120 907 samples
                            (codec/tests/test_snac.py:24-34 = 236 -> 1889 -> 15113 -> 60453 -> 120907 through strides 8, 8, 4, 2).
  * ``layers.py:123-129, 298-306``  snake(x, alpha) = x + 1 / (alpha + 1e-9) * sin(alpha x)^2, alpha stored ``[1, C, 1]``
  * ``layers.py:159-206``   Decoder: (depthwise: conv k7 groups = C, conv k1) | conv k7 -> [LocalMHA (attention.py:5-53) when attn_window_size is set: ``local_mha``]
                            -> DecoderBlocks -> snake -> conv k7 -> tanh
  * ``layers.py:209-233``   ResidualUnit: snake, conv k7 (dilation d, padding 3 d, groups), snake, conv k1, + x
  * ``layers.py:256-267``   NoiseBlock: x + noise * linear(x)  (1x1 conv, no bias); the Gaussian noise is an explicit input here.  The block reads
                            ``B, C, T = x.shape`` off a channels-LAST tensor, so ``mx.random.normal((B, 1, T))`` is [B, 1, channels]: one draw per
                            channel, constant over time (pinned by tests/test_reference_fixtures_cpu.py against the reference's own file)
  * ``layers.py:270-295``   DecoderBlock: snake, convT K = 2 s (padding ceil(s / 2)), [NoiseBlock], three units with dilations 1 / 3 / 9
  * ``vq.py:102-137``       ResidualVectorQuantize.from_codes: codebook lookup, out_proj (1x1 WNConv), repeat_interleave(stride), running sum
  * ``snac.py:104-107``     SNAC.decode(codes) = decoder(from_codes(codes).moveaxis(1, 2))

Encode side (round 5):
  * ``layers.py:132-156``   Encoder: conv k7 (1 -> d_model), EncoderBlocks (d_model doubles per block), [LocalMHA], conv k7 (groups = d_model when depthwise;
                            NO snake in front of it)
  * ``layers.py:236-253``   EncoderBlock: three units (dilations 1 / 3 / 9, groups = channels when depthwise) on the INPUT width, snake,
                            WNConv1d(K = 2 s, stride s, padding ceil(s / 2))
  * ``vq.py:10-78``         VectorQuantize: average pool over ``stride`` frames (a grouped conv with a 1 / stride kernel), in_proj, L2-normalised nearest
                            codeword (``(-dist).argmax``), out_proj of the un-normalised codeword, repeat_interleave(stride)
  * ``vq.py:102-113``       ResidualVectorQuantize.__call__: the residual loop; ``snac.py:96-102`` encode = preprocess (right pad) -> encoder -> codes
pinned the same way: ``SNAC.encode`` / ``SNAC.__call__`` of the reference's own modules on a seeded checkpoint (``ref_snac_encode.npz``; every code equal).

Parameter names are the reference's module paths (``decoder.model.layers.N...``, ``quantizer.quantizers.N.codebook.weight`` ...), layouts MLX's
(conv ``[out, K, in / groups]``, transposed conv stored ``[in, K, out]``).  Arithmetic float32 (float64 on request) on the parameters as given.
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files
(imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) and tests/test_reference_fixtures_cpu.py holds
this oracle to the result -- ``SNAC.quantizer.from_codes`` + ``SNAC.decoder`` with the NoiseBlock draws the reference made: waveform 2e-5 (this run exposed the NoiseBlock's per-channel noise).  The reference's own tests hold shape / length pins only
(reproduced in tests/test_oracle_golden.py and the GPU tests); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def wn_weight(g: Tensor, v: Tensor) -> Tensor:
    """layers.py:9-15: the norm runs over every axis but 0 -- for BOTH conv types (the transposed conv stores ``(in, K, out)``)."""
    return g * v / torch.sqrt((v ** 2).sum(dim=(1, 2), keepdim=True))


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """x [B, T, C]; alpha stored [1, C, 1] (layers.py:301-306 swaps it to [1, 1, C])."""
    a = alpha.reshape(1, 1, -1)
    return x + torch.reciprocal(a + 1e-9) * torch.sin(a * x) ** 2


class SNACDecoderRef:
    def __init__(self, weights: Dict[str, Tensor], decoder_rates: List[int], vq_strides: List[int], noise: bool = True, depthwise: bool = True,
                 dtype=torch.float32, attn_window_size: Optional[int] = None):
        self.attn_window_size = attn_window_size
        self.w = {k: v.to(dtype) if v.is_floating_point() else v for k, v in weights.items()}
        self.rates, self.vq_strides, self.noise, self.depthwise, self.dtype = list(decoder_rates), list(vq_strides), noise, depthwise, dtype

    def _conv(self, x: Tensor, name: str, dilation: int = 1, padding: int = 0, groups: int = 1) -> Tensor:
        w = wn_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # [out, K, in / groups]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), padding=padding, dilation=dilation, groups=groups).transpose(1, 2)

    def _convT(self, x: Tensor, name: str, stride: int) -> Tensor:
        w = wn_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # stored [in, K, out] = torch's [in, out, K] permuted
        return F.conv_transpose1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), stride=stride, padding=math.ceil(stride / 2),
                                  output_padding=1).transpose(1, 2)

    def local_mha(self, x: Tensor, p: str) -> Tensor:
        """PARITY UNPINNED (the reference's module raises in this layout: tests/golden/ref_snac_local_mha_probe.json) -- what ``LocalMHA.__call__``
        (attention.py:19-53) MEANS, on channels-last x [B, T, C]: LayerNorm, to_qkv (no bias), heads of 64, windows of
        ``attn_window_size`` positions, rotary embedding with the position INSIDE the window (``SinusoidalEmbeddings`` without xpos: freqs =
        [t * inv_freq | t * inv_freq], scale 1; ``apply_rotary_pos_emb``: q cos + rotate_half(q) sin, rotate_half = [-x2, x1]), scores / sqrt(64),
        softmax, to_out, + residual."""
        ws, dh = self.attn_window_size, 64
        B, T, C = x.shape
        H, W = C // dh, T // ws
        xn = F.layer_norm(x, (C,), self.w[p + "norm.weight"], self.w[p + "norm.bias"], 1e-5)
        qkv = xn @ self.w[p + "to_qkv.weight"].t()
        q, k, v = [t.reshape(B, W, ws, H, dh).permute(0, 3, 1, 2, 4) for t in qkv.split(C, dim=-1)]
        inv = 1.0 / (10000 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
        fr = torch.arange(ws, dtype=torch.float32)[:, None] * inv[None, :]
        fr = torch.cat([fr, fr], dim=-1).to(x.dtype)

        def rot(t):
            x1, x2 = t[..., : dh // 2], t[..., dh // 2:]
            return torch.cat([-x2, x1], dim=-1)

        q = q * torch.cos(fr) + rot(q) * torch.sin(fr)
        k = k * torch.cos(fr) + rot(k) * torch.sin(fr)
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        out = torch.softmax(sc, dim=-1) @ v
        out = out.permute(0, 2, 3, 1, 4).reshape(B, T, C)
        return out @ self.w[p + "to_out.weight"].t() + x

    def from_codes(self, codes: List[Tensor]) -> Tensor:
        """codes[i] int [B, T / stride_i] -> z_q [B, D, T] (vq.py:116-137)."""
        z = 0.0
        for i, c in enumerate(codes):
            p = f"quantizer.quantizers.{i}."
            e = self.w[p + "codebook.weight"][c.long()]                       # [B, T_i, d]
            zi = self._conv(e, p + "out_proj")                                 # [B, T_i, D]
            if self.vq_strides[i] > 1:
                zi = torch.repeat_interleave(zi, self.vq_strides[i], dim=1)    # expanded[..., j::stride] = z_q_i
            z = z + zi
        return z.transpose(1, 2)

    def decode(self, z: Tensor, noises: Optional[List[Tensor]] = None, return_stages: bool = False):
        """z [B, D, T] -> audio [B, T', 1]; ``noises[i]`` [B, 1, C_i] is DecoderBlock i's ``mx.random.normal((B, 1, T))`` (its "T" is the channel count)."""
        x = z.to(self.dtype).transpose(1, 2)
        st = {}
        D = x.shape[-1]
        m = "decoder.model.layers."
        if self.depthwise:
            x = self._conv(x, m + "0", padding=3, groups=D)
            x = self._conv(x, m + "1")
            nxt = 2
        else:
            x = self._conv(x, m + "0", padding=3)
            nxt = 1
        st["conv_in"] = x
        if self.attn_window_size is not None:
            x = self.local_mha(x, f"{m}{nxt}.")
            st["attn"] = x
            nxt += 1
        for i, s in enumerate(self.rates):
            p = f"{m}{nxt + i}.block.layers."
            x = snake(x, self.w[p + "0.alpha"])
            x = self._convT(x, p + "1", s)
            C = x.shape[-1]
            j0 = 2
            if self.noise:
                w = wn_weight(self.w[p + "2.linear.weight_g"], self.w[p + "2.linear.weight_v"])
                h = F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1)).transpose(1, 2)
                assert tuple(noises[i].shape) == (x.shape[0], 1, x.shape[2]), (tuple(noises[i].shape), tuple(x.shape))
                x = x + noises[i].to(self.dtype) * h
                j0 = 3
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j0 + j}.block.layers."
                y = snake(x, self.w[q + "0.alpha"])
                y = self._conv(y, q + "1", dilation=d, padding=3 * d, groups=C if self.depthwise else 1)
                y = snake(y, self.w[q + "2.alpha"])
                y = self._conv(y, q + "3")
                x = x + y
            st[f"block{i}"] = x
        n = nxt + len(self.rates)
        x = snake(x, self.w[f"{m}{n}.alpha"])
        x = torch.tanh(self._conv(x, f"{m}{n + 1}", padding=3))
        return (x, st) if return_stages else x


class SNACEncoderRef:
    """``SNAC.encode`` (snac.py:96-102): preprocess -> ``Encoder`` -> ``ResidualVectorQuantize.__call__``."""

    def __init__(self, weights: Dict[str, Tensor], encoder_rates: List[int], vq_strides: List[int], depthwise: bool = True, dtype=torch.float32,
                 attn_window_size: Optional[int] = None):
        self.w = {k: v.to(dtype) if v.is_floating_point() else v for k, v in weights.items()}
        self.rates, self.vq_strides, self.depthwise, self.dtype, self.attn_window_size = list(encoder_rates), list(vq_strides), depthwise, dtype, attn_window_size
        self._dec = SNACDecoderRef(weights, [], vq_strides, noise=False, depthwise=depthwise, dtype=dtype, attn_window_size=attn_window_size)

    def _conv(self, x: Tensor, name: str, dilation: int = 1, padding: int = 0, groups: int = 1, stride: int = 1) -> Tensor:
        w = wn_weight(self.w[name + ".weight_g"], self.w[name + ".weight_v"])  # [out, K, in / groups]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".bias"), stride=stride, padding=padding, dilation=dilation,
                        groups=groups).transpose(1, 2)

    def preprocess(self, audio: Tensor) -> Tensor:
        """snac.py:67-86 (the window size joins the least common multiple when attention is on)."""
        lcm = self.vq_strides[0]
        for s in self.vq_strides[1:]:
            lcm = abs(lcm * s) // math.gcd(lcm, s)
        if self.attn_window_size:
            lcm = abs(lcm * self.attn_window_size) // math.gcd(lcm, self.attn_window_size)
        pad_to = int(torch.tensor(self.rates).prod()) * lcm
        return F.pad(audio, (0, math.ceil(audio.shape[-1] / pad_to) * pad_to - audio.shape[-1]))

    def encoder(self, audio: Tensor, return_stages: bool = False):
        """audio [B, 1, S] -> z [B, D, T]."""
        x = audio.to(self.dtype).transpose(1, 2)
        st = {}
        e = "encoder.block.layers."
        x = self._conv(x, e + "0", padding=3)
        for i, s in enumerate(self.rates):
            p = f"{e}{i + 1}.block.layers."
            C = x.shape[-1]
            for j, d in enumerate((1, 3, 9)):
                q = p + f"{j}.block.layers."
                y = snake(x, self.w[q + "0.alpha"])
                y = self._conv(y, q + "1", dilation=d, padding=3 * d, groups=C if self.depthwise else 1)
                y = snake(y, self.w[q + "2.alpha"])
                y = self._conv(y, q + "3")
                x = x + y
            st[f"units{i}"] = x
            x = snake(x, self.w[p + "3.alpha"])
            x = self._conv(x, p + "4", stride=s, padding=math.ceil(s / 2))
            st[f"block{i}"] = x
        nxt = len(self.rates) + 1
        if self.attn_window_size is not None:
            x = self._dec.local_mha(x, f"{e}{nxt}.")
            st["attn"] = x
            nxt += 1
        x = self._conv(x, f"{e}{nxt}", padding=3, groups=x.shape[-1] if self.depthwise else 1)
        st["latent"] = x
        return (x.transpose(1, 2), st) if return_stages else x.transpose(1, 2)

    def quantize(self, z: Tensor, return_margins: bool = False):
        """z [B, D, T] -> (z_q [B, D, T], codes list [B, T / stride_i]) (+ the cosine top-2 gap per decision)."""
        residual = z.to(self.dtype)
        z_q = 0
        codes, margins = [], []
        for i, s in enumerate(self.vq_strides):
            p = f"quantizer.quantizers.{i}."
            x = residual.transpose(1, 2)                                              # [B, T, D]
            if s > 1:
                D = x.shape[2]
                x = F.conv1d(x.transpose(1, 2), torch.ones((D, 1, s), dtype=self.dtype) / s, stride=s, groups=D).transpose(1, 2)
            z_e = self._conv(x, p + "in_proj").transpose(1, 2)                        # [B, d, T / s]
            b, d, t = z_e.shape
            enc = z_e.permute(0, 2, 1).reshape(b * t, d)
            cb = self.w[p + "codebook.weight"]
            en = enc / torch.clamp(torch.sqrt((enc.abs() ** 2).sum(1, keepdim=True)), min=1e-12)
            cn = cb / torch.clamp(torch.sqrt((cb.abs() ** 2).sum(1, keepdim=True)), min=1e-12)
            dist = (en ** 2).sum(1, keepdim=True) - 2 * en @ cn.t() + (cn ** 2).sum(1, keepdim=True).t()
            top = torch.topk(-dist, 2, dim=1).values
            idx = (-dist).argmax(1).reshape(b, t)
            margins.append(((top[:, 0] - top[:, 1]) / 2).reshape(b, t))
            zq_lat = cb[idx].transpose(1, 2)
            z_q_i = self._conv((z_e + (zq_lat - z_e)).transpose(1, 2), p + "out_proj").transpose(1, 2)
            if s > 1:
                z_q_i = torch.repeat_interleave(z_q_i, s, dim=2)
            z_q = z_q + z_q_i
            residual = residual - z_q_i
            codes.append(idx)
        return (z_q, codes, margins) if return_margins else (z_q, codes)

    def encode(self, audio: Tensor):
        return self.quantize(self.encoder(self.preprocess(audio)))[1]
