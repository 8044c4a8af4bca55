"""Sortformer's feature extraction (vad/models/sortformer/sortformer.py:36-123) on the MI355X: NeMo FilterbankFeatures for a BATCH of waveforms in one
fused kernel launch.  The diarisation model itself is outside SURVEY 8(a)."""
from typing import Optional

import torch

from ....frontends import nemo_log_mel, per_feature_norm

_LOG_GUARD = 2 ** -24
_NORM_CONSTANT = 1e-5


def preemphasis_filter(waveform: torch.Tensor, coeff: float = 0.97) -> torch.Tensor:
    """y[n] = x[n] - coeff * x[n-1], first sample kept (sortformer.py:36-40)."""
    return torch.cat([waveform[..., :1], waveform[..., 1:] - coeff * waveform[..., :-1]], dim=-1)


def extract_mel_features(waveform, sample_rate: int = 16000, n_fft: int = 512, hop_length: int = 160, win_length: int = 400, n_mels: int = 80,
                         preemphasis_coeff: float = 0.97, normalize: Optional[str] = "per_feature", pad_to: int = 16) -> torch.Tensor:
    """``[num_samples]`` or ``[batch, num_samples]`` -> ``[batch, n_mels, num_frames]`` (frames zero-padded to a multiple of ``pad_to``)."""
    y = nemo_log_mel(waveform, sample_rate, n_fft, hop_length, win_length, n_mels, "hann", preemphasis_coeff, _LOG_GUARD)   # [B, frames, mels]
    feats = y.transpose(1, 2)
    if normalize == "per_feature":
        feats = per_feature_norm(feats, dim=2, eps=_NORM_CONSTANT)
    if pad_to > 0 and feats.shape[2] % pad_to:
        feats = torch.nn.functional.pad(feats, (0, pad_to - feats.shape[2] % pad_to))
    return feats.contiguous()
