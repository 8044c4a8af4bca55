"""Programme loudness and level normalisation on the host: the ``mlx_audio.dsp`` helpers ``lfilter``, ``integrated_loudness``, ``normalize_loudness``,
``normalize_peak`` (``mlx_audio/dsp.py:96-382``: numpy code in the reference too, used by its post-processing, not by the model hot path).

Restated from the published algorithm, ITU-R BS.1770-4:
  * K-weighting = a high-shelf "head" stage followed by the RLB high-pass, both given as analogue prototypes (centre frequency, Q, gain) and mapped to the
    sampling rate with the bilinear transform (the libebur128 design: reproduces Tables 1 and 2 of the Recommendation at 48 kHz to 1e-12 and keeps the
    response at any other rate; the two stages together have +0.691 dB at 997 Hz, which the -0.691 constant cancels);
  * gating blocks of 400 ms "to the nearest sample" with 75 % overlap, incomplete trailing blocks unused; block loudness
    ``-0.691 + 10 log10(sum_i G_i z_i)`` with channel weights 1, 1, 1, 1.41, 1.41; absolute gate at -70 LKFS, relative gate 10 LU below the mean of the
    blocks that passed the absolute gate; the reading is the mean over the blocks above both gates.
Signatures, defaults and validation messages follow the reference (``dsp.py:107-118, 239-244, 356-382``); the properties its tests pin (Table 1 / 2
coefficients, the 997 Hz anchor at 48 kHz and 11 025 Hz, incomplete blocks, gating, normalisation targets, the peak vector: ``tests/test_dsp.py:98-296, 381-395``)
are asserted in ``tests/test_audio_io_cpu.py``.
"""
from __future__ import annotations

import math
import warnings
from typing import Tuple

import numpy as np

_K_WEIGHT_SHELF_FREQ = 1681.974450955533
_K_WEIGHT_SHELF_Q = 0.7071752369554196
_K_WEIGHT_SHELF_GAIN_DB = 3.999843853973347
_K_WEIGHT_HIGHPASS_FREQ = 38.13547087602444
_K_WEIGHT_HIGHPASS_Q = 0.5003270373238773
_SHELF_BAND_EXPONENT = 0.4996667741545416   # Vb = Vh ** this (libebur128's fit of the BS.1770 head filter)


def _biquad_coefficients(gain_db: float, q_factor: float, center_freq: float, rate: int, filter_type: str) -> Tuple[np.ndarray, np.ndarray]:
    """(b, a) of one K-weighting stage at ``rate``; ``a[0] == 1``.  ``high_pass`` keeps the unnormalised numerator [1, -2, 1] of BS.1770 Table 2."""
    k = math.tan(math.pi * center_freq / rate)
    a0 = 1.0 + k / q_factor + k * k
    a = np.array([1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q_factor + k * k) / a0])
    if filter_type == "high_shelf":
        vh = 10.0 ** (gain_db / 20.0)
        vb = vh ** _SHELF_BAND_EXPONENT
        b = np.array([(vh + vb * k / q_factor + k * k) / a0, 2.0 * (k * k - vh) / a0, (vh - vb * k / q_factor + k * k) / a0])
    elif filter_type == "high_pass":
        b = np.array([1.0, -2.0, 1.0])
    else:
        raise ValueError(f"Unsupported filter type: {filter_type}")
    return b, a


def lfilter(b, a, data) -> np.ndarray:
    """1-D causal IIR / FIR filter, direct form II transposed: ``a[0] y[n] = b[0] x[n] + z0``, ``z_i = b[i+1] x[n] - a[i+1] y[n] + z_{i+1}``."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    x = np.asarray(data, dtype=np.float64)
    if a[0] == 0.0:
        raise ValueError("a[0] must be non-zero")
    b, a = b / a[0], a / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    if n == 1:
        return b[0] * x
    try:  # the same recurrence in C when scipy is there (it is a dependency of the resampler already)
        from scipy.signal import lfilter as _sp

        return _sp(b, a, x)
    except ImportError:  # pragma: no cover
        pass
    y = np.empty_like(x)
    z = np.zeros(n - 1)
    for i, xi in enumerate(x):
        yi = b[0] * xi + z[0]
        z[:-1] = b[1:-1] * xi - a[1:-1] * yi + z[1:]
        z[-1] = b[-1] * xi - a[-1] * yi
        y[i] = yi
    return y


def _validate_loudness_audio(data: np.ndarray, rate: int, block_size: float) -> None:
    if not isinstance(data, np.ndarray):
        raise ValueError("Data must be of type numpy.ndarray.")
    if not np.issubdtype(data.dtype, np.floating):
        raise ValueError("Data must be floating point.")
    if data.ndim == 2 and data.shape[1] > 5:
        raise ValueError("Audio must have five channels or less.")
    if data.shape[0] < block_size * rate:
        raise ValueError("Audio must have length greater than the block size.")


def _k_weight_audio(data: np.ndarray, rate: int) -> np.ndarray:
    """[samples, channels] through both K-weighting stages."""
    out = np.array(data, dtype=np.float64, copy=True)
    for gain, q, fc, kind in ((_K_WEIGHT_SHELF_GAIN_DB, _K_WEIGHT_SHELF_Q, _K_WEIGHT_SHELF_FREQ, "high_shelf"),
                              (0.0, _K_WEIGHT_HIGHPASS_Q, _K_WEIGHT_HIGHPASS_FREQ, "high_pass")):
        b, a = _biquad_coefficients(gain, q, fc, rate, kind)
        for ch in range(out.shape[1]):
            out[:, ch] = lfilter(b, a, out[:, ch])
    return out


def integrated_loudness(data: np.ndarray, rate: int, block_size: float = 0.400, overlap: float = 0.75) -> float:
    """Integrated loudness in LUFS (= LKFS) of ``data`` ``[samples]`` or ``[samples, channels <= 5]``."""
    x = np.array(data, copy=True)
    _validate_loudness_audio(x, rate, block_size)
    if x.ndim == 1:
        x = x[:, None]
    x = _k_weight_audio(x, rate)
    gains = np.array([1.0, 1.0, 1.0, 1.41, 1.41])[: x.shape[1]]
    block = int(round(block_size * rate))
    hop = max(1, int(round(block_size * (1.0 - overlap) * rate)))
    n_blocks = (x.shape[0] - block) // hop + 1 if x.shape[0] >= block else 0
    if n_blocks <= 0:
        raise ValueError("Audio must have length greater than the block size.")
    sq = np.concatenate([np.zeros((1, x.shape[1])), np.cumsum(x * x, axis=0)], axis=0)     # prefix sums: a block's energy is one subtraction
    starts = np.arange(n_blocks) * hop
    z = (sq[starts + block] - sq[starts]) / block                                             # mean square [blocks, channels]
    weighted = z @ gains
    with np.errstate(divide="ignore"):
        block_lufs = -0.691 + 10.0 * np.log10(weighted)
        keep = block_lufs > -70.0
        if not keep.any():
            return float("-inf")
        relative = -0.691 + 10.0 * np.log10(weighted[keep].mean()) - 10.0
        keep &= block_lufs > relative
        if not keep.any():
            return float("-inf")
        return float(-0.691 + 10.0 * np.log10(weighted[keep].mean()))


def normalize_loudness(data: np.ndarray, input_loudness: float, target_loudness: float) -> np.ndarray:
    """Scales ``data`` by the gain that moves ``input_loudness`` to ``target_loudness`` (LUFS); warns when the result may clip."""
    out = np.power(10.0, (target_loudness - input_loudness) / 20.0) * data
    if np.max(np.abs(out)) >= 1.0:
        warnings.warn("Possible clipped samples in output.")
    return out


def normalize_peak(data: np.ndarray, target_peak_db: float) -> np.ndarray:
    """Scales ``data`` so that its absolute peak sits at ``target_peak_db`` dBFS; warns when the result may clip."""
    out = (np.power(10.0, target_peak_db / 20.0) / np.max(np.abs(data))) * data
    if np.max(np.abs(out)) >= 1.0:
        warnings.warn("Possible clipped samples in output.")
    return out
