"""PyTorch-CPU / numpy restatement of the Vocos vocoder (TEST ORACLE, not product).

Follows /root/reference/mlx_audio/codec/models/vocos statement by statement:
  * ``mel.py:9-33``            log_mel_spectrogram: centre-reflect STFT (n_fft 1024, hop n_fft // 4 because ``win_length=`` is what receives
                               the caller's hop and an array window ignores it -- sic), last frame dropped, |X| @ htk filters (no norm),
                               log(max(., 1e-5)), leading batch axis
  * ``vocos.py:137-190``       ConvNeXtBlock: depthwise k7 conv, LayerNorm / AdaLayerNorm(eps 1e-6), Linear, exact GELU, Linear, gamma, residual
  * ``vocos.py:193-211``       AdaLayerNorm: affine-free layer norm, then * scale(cond) + shift(cond) with Linear(num_embeddings -> dim) on the
                               conditioning VECTOR (the reference feeds ``bandwidth_id`` [1, n] through a Linear, not an Embedding -- sic)
  * ``vocos.py:214-273``       VocosBackbone: embed conv k7, norm, blocks, final LayerNorm(eps 1e-6)
  * ``vocos.py:116-134``       ISTFTHead: Linear(dim, n_fft + 2), split, exp, clip(max 1e2), cos / sin, dsp.istft(window=hanning(n_fft)
                               symmetric array, plain-window overlap-add normalisation, centre trim); ``padding`` is accepted and unused (sic)

Parameter names are the reference's after its own load-time transposes (``vocos.py:337-347``): ``backbone.embed.weight`` [dim, K, Cin],
``backbone.convnext.{i}.dwconv.weight`` [dim, K, 1], ``...pwconv1.weight`` [inter, dim], ``...gamma`` [dim], ``head.out.weight`` [n_fft+2, dim].
Arithmetic float32 (float64 on request) on the parameters as given (the published checkpoints are float32).

Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files
(imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) and tests/test_reference_fixtures_cpu.py holds
this oracle to the result -- mel ``Vocos``: features 2e-5, waveform 5e-5.  The reference's own tests hold shape / length pins only
(reproduced in tests/test_oracle_golden.py and the GPU tests); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import dsp_ref

Tensor = torch.Tensor


def log_mel_spectrogram(audio, sample_rate: int = 24_000, n_mels: int = 100, n_fft: int = 1024, hop_length: int = 256, padding: int = 0) -> np.ndarray:
    """mel.py:9-33 -> [1, n_frames - 1, n_mels] float32."""
    audio = np.asarray(audio, dtype=np.float32)
    if padding > 0:
        audio = np.pad(audio, (0, padding))
    freqs = dsp_ref.stft(audio, window=dsp_ref.hanning(n_fft), n_fft=n_fft, win_length=hop_length)  # hop = n_fft // 4 (sic)
    mag = np.abs(freqs[:-1, :]).astype(np.float32)
    fb = dsp_ref.mel_filters(sample_rate, n_fft, n_mels, norm=None, mel_scale="htk")
    mel = (mag.astype(np.float64) @ fb.T.astype(np.float64)).astype(np.float32)
    return np.log(np.maximum(mel, np.float32(1e-5)))[None]


class VocosRef:
    def __init__(self, weights: Dict[str, Tensor], config: dict, dtype=torch.float32):
        self.cfg = config
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        b = config["backbone"]["init_args"]
        self.input_channels, self.dim, self.num_layers = b["input_channels"], b["dim"], b["num_layers"]
        self.adanorm = b.get("adanorm_num_embeddings") is not None
        h = config["head"]["init_args"]
        self.n_fft, self.hop = h["n_fft"], h["hop_length"]

    def _norm(self, x: Tensor, name: str, cond: Optional[Tensor]) -> Tensor:
        if self.adanorm and (name + ".scale.weight") in self.w:
            scale = F.linear(cond, self.w[name + ".scale.weight"], self.w[name + ".scale.bias"])
            shift = F.linear(cond, self.w[name + ".shift.weight"], self.w[name + ".shift.bias"])
            xh = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)
            return xh * scale[:, None, :] + shift[:, None, :]
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w.get(name + ".bias"), 1e-6)

    def backbone(self, x: Tensor, bandwidth_id: Optional[Tensor] = None, return_layers: bool = False):
        """x [B, T, Cin] (or [B, Cin, T], transposed like vocos.py:253-255) -> [B, T, dim]."""
        x = x.to(self.dtype)
        cond = None if bandwidth_id is None else bandwidth_id.to(self.dtype)
        if x.shape[-1] != self.input_channels:
            x = x.transpose(1, 2)
        w = self.w["backbone.embed.weight"]  # [dim, K, Cin]
        x = F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w["backbone.embed.bias"], padding=w.shape[1] // 2).transpose(1, 2)
        x = self._norm(x, "backbone.norm", cond)
        layers = [x]
        for i in range(self.num_layers):
            p = f"backbone.convnext.{i}."
            dw = self.w[p + "dwconv.weight"]  # [dim, K, 1]
            r = x
            x = F.conv1d(x.transpose(1, 2), dw.permute(0, 2, 1), self.w[p + "dwconv.bias"], padding=dw.shape[1] // 2, groups=self.dim).transpose(1, 2)
            x = self._norm(x, p + "norm", cond)
            x = F.gelu(F.linear(x, self.w[p + "pwconv1.weight"], self.w[p + "pwconv1.bias"]))
            x = F.linear(x, self.w[p + "pwconv2.weight"], self.w[p + "pwconv2.bias"])
            if (p + "gamma") in self.w:
                x = self.w[p + "gamma"] * x
            x = r + x
            layers.append(x)
        x = F.layer_norm(x, (self.dim,), self.w["backbone.final_layer_norm.weight"], self.w.get("backbone.final_layer_norm.bias"), 1e-6)
        return (x, layers) if return_layers else x

    def head(self, x: Tensor, return_spec: bool = False):
        """x [1, T, dim] -> audio [(T - 1) * hop] (vocos.py:126-134; ``S.squeeze(0)``: batch of one)."""
        y = F.linear(x, self.w["head.out.weight"], self.w["head.out.bias"]).transpose(1, 2)  # [1, n_fft + 2, T]
        mag, p = y.split(y.shape[1] // 2, dim=1)
        mag = torch.clamp(torch.exp(mag), max=1e2)
        S = (mag * torch.cos(p)).double().numpy() + 1j * (mag * torch.sin(p)).double().numpy()
        audio = dsp_ref.istft(S[0], window=dsp_ref.hanning(self.n_fft), hop_length=self.hop, win_length=self.n_fft)
        return (audio, S[0]) if return_spec else audio

    def decode(self, features, bandwidth_id: Optional[Tensor] = None) -> np.ndarray:
        x = self.backbone(torch.as_tensor(features), bandwidth_id)
        return self.head(x)

    def __call__(self, audio, bandwidth_id: Optional[Tensor] = None) -> np.ndarray:
        fe = self.cfg["feature_extractor"]["init_args"]
        feats = log_mel_spectrogram(audio, sample_rate=fe.get("sample_rate", 24000), n_mels=fe.get("n_mels", 100), n_fft=fe.get("n_fft", 1024),
                                    hop_length=fe.get("hop_length", 256), padding=0)
        return self.decode(feats, bandwidth_id)
