"""Sample-rate conversion at the edge of the hot path: ``resample_audio_array`` (``mlx_audio/resample.py:29-47``) and ``utils.resample_audio``
(``mlx_audio/utils.py:541-578``) with the reference's filter, on the host.

The reference's resampler is host code too (scipy): a polyphase FIR whose anti-aliasing filter is the resampy ``kaiser_best`` design -- a Kaiser-windowed
sinc with 64 zero crossings per side, roll-off 0.9475937167399596 and beta 14.769656459379492, designed at the upsampled rate with its cut-off at
``rolloff / max(up, down)`` of that Nyquist (``resample.py:15-26``) -- applied by ``scipy.signal.resample_poly`` with edge padding (``:40-47``).  The
design constants ARE the algorithm; they are restated here and the properties the reference's tests pin (energy above the new Nyquist removed,
pass band at unit gain, length, identity at equal rates: ``mlx_audio/tests/test_dsp.py:299-349``) are asserted in ``tests/test_audio_io_cpu.py``.
A device-side polyphase kernel is the SURVEY 8(f).4 follow-up; the chunked variant (``resample_audio_chunks``) is not built.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

KAISER_BEST_ZEROS = 64
KAISER_BEST_ROLLOFF = 0.9475937167399596
KAISER_BEST_BETA = 14.769656459379492


@lru_cache(maxsize=32)
def polyphase_design(orig_sample_rate: int, sample_rate: int):
    """(up, down, FIR taps) of the conversion ``orig_sample_rate`` -> ``sample_rate``."""
    from scipy import signal

    g = math.gcd(int(orig_sample_rate), int(sample_rate))
    up, down = int(sample_rate) // g, int(orig_sample_rate) // g
    widest = max(up, down)
    taps = signal.firwin(2 * KAISER_BEST_ZEROS * widest + 1, KAISER_BEST_ROLLOFF / widest, window=("kaiser", KAISER_BEST_BETA))
    return up, down, taps


def resample_audio_array(audio: np.ndarray, orig_sample_rate: int, sample_rate: int, axis: int = -1) -> np.ndarray:
    """In-memory array through the polyphase FIR; float32 out, the input itself when the rates are equal."""
    if orig_sample_rate == sample_rate:
        return audio
    from scipy import signal

    up, down, taps = polyphase_design(int(orig_sample_rate), int(sample_rate))
    return signal.resample_poly(np.asarray(audio), up, down, axis=axis, window=taps, padtype="edge").astype(np.float32, copy=False)


def resample_audio(audio, orig_sample_rate: int, sample_rate: int, axis: int = -1):
    """``mlx_audio.utils.resample_audio``: numpy in -> numpy out, torch tensor in -> torch tensor out (same device), identity at equal rates."""
    if orig_sample_rate == sample_rate:
        return audio
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(audio, torch.Tensor):
        out = resample_audio_array(audio.detach().cpu().numpy(), orig_sample_rate, sample_rate, axis=axis)
        return torch.from_numpy(np.ascontiguousarray(out)).to(audio.device)
    return resample_audio_array(np.asarray(audio), orig_sample_rate, sample_rate, axis=axis)
