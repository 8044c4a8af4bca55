"""numpy restatement of ``tts/models/interpolate.py:61-117`` (TEST ORACLE, not product).

Index arithmetic is done in float32 exactly as the reference's MLX ops do it
(``mx.arange(size)`` int32 times a Python-float scalar -> float32, one rounding
per op); this matters for SineGen, whose x300 up-sampling multiplies index
rounding errors by phase slopes of hundreds of radians.

Parity status: **pinned** to the reference's own interpolate vectors (``tts/tests/test_interpolate.py:40-97``: nearest / linear, align_corners on and
off, scale factors and sizes; transcribed into tests/golden/reference_vectors.json, tests/test_oracle_golden.py::test_interpolate_golden); the
reference's ``interpolate.py`` itself, executed over the numpy stand-in for MLX, reproduces the same vectors (``check_shim_against_reference_vectors``
in tests/golden/make_reference_fixtures.py) and runs inside every Kokoro / KittenTTS reference fixture this oracle's callers are held to.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

F32 = np.float32


def output_size(in_width: int, size=None, scale_factor=None) -> int:
    """interpolate.py:26-50: ``max(1, ceil(float(W) * float(scale)))`` in Python doubles."""
    if (size is None) == (scale_factor is None):
        raise ValueError("exactly one of size / scale_factor must be given")
    if size is not None:
        return int(size)
    return max(1, int(math.ceil(float(in_width) * float(scale_factor))))


def linear_coords(in_width: int, size: int, align_corners: Optional[bool]):
    """Source coordinate, low/high index and fraction, all as the reference computes them."""
    i = np.arange(size).astype(F32)
    if align_corners and size > 1:
        x = (i * F32((in_width - 1) / (size - 1))).astype(F32)
    elif size == 1:
        x = np.zeros(1, F32)
    else:
        x = (i * F32(in_width / size)).astype(F32)
        if not align_corners:
            x = (x + F32(0.5 * (in_width / size))).astype(F32)
            x = (x - F32(0.5)).astype(F32)
            x = np.maximum(x, F32(0.0))
    lo = np.floor(x).astype(np.int32)
    hi = np.minimum(lo + 1, in_width - 1)
    frac = (x - lo.astype(F32)).astype(F32)
    return lo, hi, frac


def interpolate1d(inp, size: int, mode: str = "linear", align_corners: Optional[bool] = None):
    """``[N, C, W] -> [N, C, size]``; nearest = floor(i*W/size), linear = torch semantics."""
    inp = np.asarray(inp, dtype=F32)
    n, c, w = inp.shape
    size = max(int(size), 1)
    if mode == "nearest":
        if size == 1:
            idx = np.zeros(1, np.int32)
        else:
            idx = np.floor(np.arange(size).astype(F32) * F32(w / size)).astype(np.int32)
            idx = np.clip(idx, 0, w - 1)
        return inp[:, :, idx]
    if w == 1:
        return np.broadcast_to(inp, (n, c, size)).copy()
    lo, hi, frac = linear_coords(w, size, align_corners)
    one_minus = (F32(1) - frac).astype(F32)
    a = (inp[:, :, lo] * one_minus[None, None, :]).astype(F32)
    b = (inp[:, :, hi] * frac[None, None, :]).astype(F32)
    return (a + b).astype(F32)


def interpolate(inp, size=None, scale_factor=None, mode="nearest", align_corners=None):
    inp = np.asarray(inp)
    if inp.ndim < 3:
        raise ValueError(f"Expected at least 3D input (N, C, D1), got {inp.ndim}D")
    if size is not None and scale_factor is not None:
        raise ValueError("Only one of size or scale_factor should be defined")
    if size is None and scale_factor is None:
        raise ValueError("One of size or scale_factor must be defined")
    if inp.ndim != 3:
        raise ValueError(f"Only 1D interpolation currently supported, got {inp.ndim - 2}D")
    if isinstance(size, (list, tuple)):
        size = size[0]
    if isinstance(scale_factor, (list, tuple)):
        scale_factor = scale_factor[0]
    return interpolate1d(inp, output_size(inp.shape[2], size, scale_factor), mode, align_corners)
