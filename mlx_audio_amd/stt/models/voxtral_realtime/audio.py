"""Voxtral Realtime's mel front end (stt/models/voxtral_realtime/audio.py): periodic Hann(400), hop 160, reflect-centred, last frame dropped,
Slaney filters 0-8000 Hz, log10 clamped to the FIXED maximum 1.5 - 8, (. + 4) / 4 -- one fused kernel + the clamp pass."""
import numpy as np
import torch

from .... import dsp
from ....frontends import whisper_style_log_mel


def compute_mel_filters(num_mel_bins: int = 128, window_size: int = 400, sample_rate: int = 16000) -> np.ndarray:
    """``[n_fft/2+1, num_mel_bins]`` float32 (audio.py:21-38)."""
    return np.ascontiguousarray(dsp.mel_filters(sample_rate, window_size, num_mel_bins, 0, 8000, norm="slaney", mel_scale="slaney").numpy().T)


def compute_mel_spectrogram(audio, mel_filters=None, window_size: int = 400, hop_length: int = 160, global_log_mel_max: float = 1.5) -> torch.Tensor:
    """``[L]`` samples -> ``[mel_bins, n_frames - 1]`` (audio.py:41-96).  ``mel_filters`` is accepted for signature parity; the device copy of the same
    Slaney bank (0-8000 Hz, its column count = ``mel_filters.shape[1]`` when given) is what the kernel reads."""
    n_mels = 128 if mel_filters is None else int(mel_filters.shape[1])
    if mel_filters is not None:   # a bank other than the built-in one (say, loaded from a checkpoint) must not be silently replaced
        own = compute_mel_filters(n_mels, window_size, 16000)
        given = np.asarray(mel_filters.detach().cpu().float().numpy() if hasattr(mel_filters, "detach") else mel_filters, dtype=np.float32)
        if given.shape != own.shape or not np.allclose(given, own, rtol=1e-4, atol=1e-6):
            raise ValueError("compute_mel_spectrogram: mel_filters differs from the built-in 16 kHz / 0-8000 Hz Slaney bank the device kernel applies")
    y = whisper_style_log_mel(audio, 16000, window_size, hop_length, n_mels, periodic_window=True, drop_last=True, f_max=8000, fixed_max=global_log_mel_max)
    return y[0].t().contiguous()
