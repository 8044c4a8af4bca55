"""Vocos mel front end on the GPU (``codec/models/vocos/mel.py:9-33``): one fused launch STFT -> |X| -> htk mel -> log(max(., 1e-5))."""
from __future__ import annotations

import torch

from .... import dsp, ops


def log_mel_spectrogram(audio, sample_rate: int = 24_000, n_mels: int = 100, n_fft: int = 1024, hop_length: int = 256, padding: int = 0) -> torch.Tensor:
    """``[L]`` -> ``[1, n_frames - 1, n_mels]``.  The reference passes ``win_length=hop_length`` to ``stft`` with an ARRAY window, so the
    hop it actually runs is ``n_fft // 4`` and ``hop_length`` never reaches the transform (mel.py:22, dsp.py:394-405); preserved."""
    dev = dsp._device()
    x = torch.as_tensor(audio, dtype=torch.float32).to(dev)
    if x.dim() != 1:
        raise ValueError("vocos log_mel_spectrogram takes a 1-D signal (the reference's stft does)")
    if padding > 0:
        x = torch.nn.functional.pad(x, (0, padding))
    hop = n_fft // 4
    L = x.numel()
    if L <= n_fft // 2:
        raise ValueError(f"Input is too short (length={L + 2 * (n_fft // 2)}) for n_fft={n_fft} with hop_length={hop} and center=True.")
    n_frames = 1 + L // hop
    if n_frames < 2:
        raise ValueError("vocos log_mel_spectrogram: fewer than two STFT frames (the reference drops the last one)")
    win = dsp.hanning(n_fft).to(dev)
    fb = dsp.mel_filters(sample_rate, n_fft, n_mels, norm=None, mel_scale="htk").to(dev).contiguous()
    return ops.logmel(x[None].contiguous(), n_fft, hop, win, 1, n_frames - 1, fb, 3)
