"""Audio constants and helpers of ``mlx_audio/stt/models/whisper/audio.py`` (log-mel itself: ``mlx_audio_amd.dsp``)."""
from __future__ import annotations

import torch

from ....dsp import log_mel_spectrogram  # noqa: F401  (audio.py:41-82, fused STFT -> |X|^2 -> mel -> log10 kernel)

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN


def pad_or_trim(array: torch.Tensor, length: int = N_SAMPLES, *, axis: int = -1) -> torch.Tensor:
    """audio.py:23-38."""
    if array.shape[axis] > length:
        array = array.narrow(axis, 0, length)
    if array.shape[axis] < length:
        pad = [0, 0] * array.dim()
        ax = axis % array.dim()
        pad[2 * (array.dim() - 1 - ax) + 1] = length - array.shape[axis]
        array = torch.nn.functional.pad(array, pad)
    return array
