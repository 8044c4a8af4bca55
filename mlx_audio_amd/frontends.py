"""Log-mel front ends of more ``mlx_audio.dsp`` callers on the fused STFT -> power -> mel -> log kernel (csrc/fft_fast.h):
the shared device-side pieces.  The reference-named entry points live where the reference keeps them:

    mlx_audio_amd/stt/models/parakeet/audio.py            log_mel_spectrogram(x, PreprocessArgs)        (stt/models/parakeet/audio.py:39-94)
    mlx_audio_amd/vad/models/sortformer/sortformer.py     extract_mel_features(waveform, ...)           (vad/models/sortformer/sortformer.py:36-123)
    mlx_audio_amd/codec/models/s3/utils.py                log_mel_spectrogram(audio, ...)               (codec/models/s3/utils.py:8-42)
    mlx_audio_amd/stt/models/voxtral_realtime/audio.py    compute_mel_spectrogram(audio, mel_filters)   (stt/models/voxtral_realtime/audio.py:41-96)

Everything returns float32 torch tensors on the ROCm device; a missing GPU / library raises (no CPU fallback).
"""
from functools import lru_cache
from typing import Optional

import numpy as np
import torch

from . import dsp


@lru_cache(maxsize=32)
def _consts(sample_rate: int, n_fft: int, win_length: int, n_mels: int, window: str, periodic: bool, centre_pad: bool, f_min: float, f_max: Optional[float],
            dev_index: int):
    """(window [n_fft], filterbank [n_mels, n_fft/2+1]) on the device, built once per configuration."""
    fn = dsp.STR_TO_WINDOW_FN.get(window, dsp.hanning)
    w = fn(win_length + 1)[:-1] if periodic else fn(win_length)
    w = torch.as_tensor(w, dtype=torch.float32)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2 if centre_pad else 0       # torch.stft / NeMo centre the window in the frame; dsp.stft right-pads it
        w = torch.cat([torch.zeros(left), w, torch.zeros(n_fft - win_length - left)])
    fb = dsp.mel_filters(sample_rate, n_fft, n_mels, f_min, f_max, norm="slaney", mel_scale="slaney")
    dev = torch.device("cuda", dev_index)
    return w.to(dev).contiguous(), fb.to(dev).contiguous()


def _as_batch(x) -> torch.Tensor:
    x = torch.as_tensor(x, dtype=torch.float32).to(dsp._device())
    return (x if x.dim() == 2 else x[None]).contiguous()


def nemo_log_mel(x, sample_rate: int = 16000, n_fft: int = 512, hop_length: int = 160, win_length: int = 400, n_mels: int = 80, window: str = "hann",
                 preemph: float = 0.97, log_guard: float = 2.0 ** -24) -> torch.Tensor:
    """NeMo's FilterbankFeatures chain, batched: pre-emphasis (first sample kept) -> centred zero padding -> centre-padded window -> |X|^2 -> Slaney
    mel -> ln(mel + guard): ``[B, L]`` (or ``[L]``) -> ``[B, n_frames, n_mels]``.  One elementwise pass for the pre-emphasis, one fused kernel for the rest."""
    from . import ops

    x = _as_batch(x)
    if preemph > 0:
        x = torch.cat([x[:, :1], x[:, 1:] - preemph * x[:, :-1]], dim=1).contiguous()
    w, fb = _consts(sample_rate, n_fft, win_length, n_mels, window, False, True, 0.0, None, x.device.index or 0)
    n_frames = 1 + x.shape[1] // hop_length     # centred: 1 + (L + 2 (n_fft / 2) - n_fft) / hop
    return ops.logmel(x, n_fft, hop_length, w, 2, n_frames, fb, 4, log_guard=log_guard)


def per_feature_norm(y: torch.Tensor, dim: int, eps: float = 1e-5) -> torch.Tensor:
    """(y - mean) / (std + eps) along ``dim`` with Bessel's correction (parakeet/audio.py:80-85, sortformer.py:105-112)."""
    mean = y.mean(dim=dim, keepdim=True)
    n = max(y.shape[dim] - 1, 1)
    var = ((y - mean) ** 2).sum(dim=dim, keepdim=True) / n
    return (y - mean) / (var.sqrt() + eps)


def whisper_style_log_mel(x, sample_rate: int, n_fft: int, hop_length: int, n_mels: int, periodic_window: bool, drop_last: bool, f_max: Optional[float] = None,
                          fixed_max: Optional[float] = None) -> torch.Tensor:
    """Whisper's chain with the variations its descendants use (periodic window, last frame kept or dropped, fixed clamp maximum):
    ``[B, L]`` (or ``[L]``) -> ``[B, n_frames, n_mels]`` = (max(log10(max(mel, 1e-10)), M - 8) + 4) / 4 with M the global maximum of each item or ``fixed_max``."""
    from . import ops

    x = _as_batch(x)
    w, fb = _consts(sample_rate, n_fft, n_fft, n_mels, "hann", periodic_window, False, 0.0, f_max, x.device.index or 0)
    n_frames = 1 + x.shape[1] // hop_length - (1 if drop_last else 0)
    return ops.logmel(x, n_fft, hop_length, w, 1, n_frames, fb, 0, fixed_max=fixed_max)
