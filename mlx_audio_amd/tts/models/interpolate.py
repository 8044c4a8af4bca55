"""``interpolate`` / ``interpolate1d`` of the reference (``mlx_audio/tts/models/interpolate.py:7-132``: nearest, and linear with torch semantics)
on MI355X: same names, arguments and error behaviour; tensors are float32 torch tensors ``[N, C, W]`` on the ROCm device and the gather /
blend runs in ``mi355_interpolate1d`` (csrc/glue.hip).  Kokoro's SineGen uses the same arithmetic fused into its own kernel (csrc/source.hip);
this is the standalone operator.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from ... import _lib, ops


def interpolate(input: torch.Tensor, size: Optional[Union[int, Tuple[int, ...], List[int]]] = None,
                scale_factor: Optional[Union[float, List[float], Tuple[float, ...]]] = None, mode: str = "nearest",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    ndim = input.dim()
    if ndim < 3:
        raise ValueError(f"Expected at least 3D input (N, C, D1), got {ndim}D")
    spatial_dims = ndim - 2
    if size is not None and scale_factor is not None:
        raise ValueError("Only one of size or scale_factor should be defined")
    elif size is None and scale_factor is None:
        raise ValueError("One of size or scale_factor must be defined")
    if size is not None and not isinstance(size, (list, tuple)):
        size = [size] * spatial_dims
    if scale_factor is not None and not isinstance(scale_factor, (list, tuple)):
        scale_factor = [scale_factor] * spatial_dims
    if size is None:
        size = [max(1, int(math.ceil(float(input.shape[i + 2]) * float(scale_factor[i])))) for i in range(spatial_dims)]
    if spatial_dims == 1:
        return interpolate1d(input, size[0], mode, align_corners)
    raise ValueError(f"Only 1D interpolation currently supported, got {spatial_dims}D")


def interpolate1d(input: torch.Tensor, size: int, mode: str = "linear", align_corners: Optional[bool] = None) -> torch.Tensor:
    ops.require_gpu()
    x = input.to(device="cuda", dtype=torch.float32) if not input.is_cuda else input.to(torch.float32)
    n, c, w = x.shape
    size = max(int(size), 1)
    w_eff = max(w, 1)
    x2 = x.reshape(n * c, w).contiguous()
    y = torch.empty((n * c, size), dtype=torch.float32, device=x.device)
    align = bool(align_corners)
    if mode != "nearest" and align and size > 1:
        scale = np.float32((w_eff - 1) / (size - 1))
    else:
        scale = np.float32(w_eff / size)
    half = np.float32(0.5 * (w_eff / size))
    _lib.call_struct("mi355_interpolate1d", "mi355_interp1d_args", ops._stream(), x=x2.data_ptr(), x_rstride=x2.stride(0), W=w_eff, rows=n * c,
                     size=size, mode=0 if mode == "nearest" else 1, align_corners=int(align), scale=float(scale), half_scale=float(half),
                     y=y.data_ptr(), y_rstride=y.stride(0))
    return y.reshape(n, c, size)
