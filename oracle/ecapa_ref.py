"""PyTorch-CPU restatement of the Qwen3-TTS ECAPA-TDNN speaker encoder (TEST ORACLE, not product).

Follows ``tts/models/qwen3_tts/speaker_encoder.py`` of the reference:
  * ``:11-26``    reflect_pad_1d: mirror without repeating the edge sample, on the time axis
  * ``:29-58``    TimeDelayNetBlock: reflect pad (K - 1) * dil // 2 on each side -> conv1d (weights (out, K, in)) -> ReLU
  * ``:61-105``   Res2NetBlock: ``scale`` channel chunks; chunk 0 passes through, chunk 1 -> block 0, chunk i >= 2 -> block i - 1 of (chunk + previous output)
  * ``:108-141``  SqueezeExcitationBlock: time mean -> conv k1 -> ReLU -> conv k1 -> sigmoid -> x * gate
  * ``:144-180``  SqueezeExcitationRes2NetBlock: tdnn1 -> res2net -> tdnn2 -> SE, + the block input
  * ``:183-229``  AttentiveStatisticsPooling: [x, mean, std] (std = sqrt(biased var + 1e-12)) -> TDNN k1 -> tanh -> conv k1 -> softmax over time ->
                  weighted mean, sqrt(clip(weighted var, 1e-12)) -> [mean ; std]
  * ``:232-313``  Qwen3TTSSpeakerEncoder: TDNN(mel) -> SE-Res2Net blocks -> concatenation of the SE-Res2Net outputs -> mfa TDNN -> ASP -> fc (conv k1)
Tensors are kept time-major ``[B, T, C]`` (the layout ``nn.Conv1d`` of MLX takes); the reference transposes to ``[B, C, T]`` between layers,
which changes no value.  ``dtype=torch.float64`` gives the reference arithmetic at double precision for tolerance budgeting.
Parity status: **pinned to the reference's own module** (round 3): tests/golden/make_reference_fixtures.py ``run_qwen3_speaker_encoder`` executes
the reference's ``Qwen3TTSSpeakerEncoder`` (imported from /root/reference, unmodified, over the numpy stand-in for MLX) on a seeded checkpoint and a
seeded mel; tests/test_reference_fixtures_cpu.py holds this restatement to its embedding (1e-5 of the peak).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def reflect_pad_time(x: Tensor, pad: int) -> Tensor:
    """speaker_encoder.py:11-26 on ``[B, T, C]``."""
    if pad <= 0:
        return x
    left = x[:, 1:pad + 1].flip(1)
    right = x[:, -(pad + 1):-1].flip(1)
    return torch.cat([left, x, right], 1)


def conv_nlc(x: Tensor, w: Tensor, b: Tensor, dil: int = 1) -> Tensor:
    """``nn.Conv1d`` of MLX, no padding: x [B, T, Cin], w (Cout, K, Cin) -> [B, T - (K - 1) * dil, Cout]."""
    return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), b, dilation=dil).transpose(1, 2)


class EcapaRef:
    def __init__(self, weights: Dict[str, Tensor], cfg, dtype=torch.float32):
        """``weights``: the reference's parameter names without the ``speaker_encoder.`` prefix (what its ``sanitize`` returns); ``cfg``: any object
        with the fields of ``Qwen3TTSSpeakerEncoderConfig`` (config.py:8-33)."""
        self.w = {k: v.detach().to(dtype) for k, v in weights.items()}
        self.cfg = cfg
        self.dtype = dtype

    def _tdnn(self, x: Tensor, name: str, k: int, dil: int) -> Tensor:
        return torch.relu(conv_nlc(reflect_pad_time(x, (k - 1) * dil // 2), self.w[f"{name}.conv.weight"], self.w[f"{name}.conv.bias"], dil))

    def _res2net(self, x: Tensor, name: str, k: int, dil: int) -> Tensor:
        chunks = torch.chunk(x, self.cfg.enc_res2net_scale, dim=2)
        outs, part = [], None
        for i, c in enumerate(chunks):
            if i == 0:
                part = c
            elif i == 1:
                part = self._tdnn(c, f"{name}.blocks.{i - 1}", k, dil)
            else:
                part = self._tdnn(c + part, f"{name}.blocks.{i - 1}", k, dil)
            outs.append(part)
        return torch.cat(outs, 2)

    def _se(self, x: Tensor, name: str) -> Tensor:
        m = x.mean(1, keepdim=True)
        g = torch.relu(conv_nlc(m, self.w[f"{name}.conv1.weight"], self.w[f"{name}.conv1.bias"]))
        g = torch.sigmoid(conv_nlc(g, self.w[f"{name}.conv2.weight"], self.w[f"{name}.conv2.bias"]))
        return x * g

    def _block(self, x: Tensor, name: str, k: int, dil: int) -> Tensor:
        h = self._tdnn(x, f"{name}.tdnn1", 1, 1)
        h = self._res2net(h, f"{name}.res2net_block", k, dil)
        h = self._tdnn(h, f"{name}.tdnn2", 1, 1)
        return self._se(h, f"{name}.se_block") + x

    def _asp(self, x: Tensor, stages=None) -> Tensor:
        eps = 1e-12
        T = x.shape[1]
        mean = x.mean(1, keepdim=True)
        std = torch.sqrt(x.var(1, unbiased=False, keepdim=True) + eps)
        att = torch.cat([x, mean.expand(-1, T, -1), std.expand(-1, T, -1)], 2)
        att = torch.tanh(self._tdnn(att, "asp.tdnn", 1, 1))
        att = conv_nlc(att, self.w["asp.conv.weight"], self.w["asp.conv.bias"])
        if stages is not None:
            stages["asp_logits"] = att
        att = torch.softmax(att, 1)
        mean = (att * x).sum(1, keepdim=True)
        var = (att * (x - mean) ** 2).sum(1, keepdim=True)
        std = torch.sqrt(var.clamp_min(eps))
        return torch.cat([mean, std], 2)  # [B, 1, 2C]

    def __call__(self, mels: Tensor, stages=None) -> Tensor:
        """mels [B, T, mel_dim] -> speaker embedding [B, enc_dim]."""
        c = self.cfg
        x = mels.to(self.dtype)
        x = self._tdnn(x, "blocks.0", c.enc_kernel_sizes[0], c.enc_dilations[0])
        hs = []
        for i in range(1, len(c.enc_channels) - 1):
            x = self._block(x, f"blocks.{i}", c.enc_kernel_sizes[i], c.enc_dilations[i])
            hs.append(x)
            if stages is not None:
                stages[f"block{i}"] = x
        x = self._tdnn(torch.cat(hs, 2), "mfa", c.enc_kernel_sizes[-1], c.enc_dilations[-1])
        if stages is not None:
            stages["mfa"] = x
        x = self._asp(x, stages)
        if stages is not None:
            stages["pooled"] = x
        return conv_nlc(x, self.w["fc.weight"], self.w["fc.bias"])[:, 0]
