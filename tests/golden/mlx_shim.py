"""A numpy stand-in for the parts of ``mlx.core`` / ``mlx.nn`` the reference's Kokoro / KittenTTS modules use -- TEST INFRASTRUCTURE ONLY.

Why it exists: MLX (0.31.2, an un-vendored dependency of the reference) cannot be installed here or on the GPU box, so the reference cannot run
as-is.  With this stand-in registered under the names ``mlx`` / ``mlx.core`` / ``mlx.nn`` the reference's OWN source files
(``/root/reference/mlx_audio/tts/models/{kokoro,kitten_tts}/*.py``, ``dsp.py``, ``interpolate.py``) are imported from where they lie and executed
unmodified on the CPU; ``tests/golden/make_reference_fixtures.py`` does that and stores what the reference's code computes as fixtures the oracle
is then pinned to (``tests/test_reference_fixtures_cpu.py``).  What that pins: the reference's composition -- module wiring, slicing, transposes,
padding, masks, the order of operations, parameter names -- i.e. everything ``oracle/*.py`` restates by hand.  What it cannot pin: MLX's own
kernels (each primitive here follows MLX's documented semantics: channels-last convolutions with ``(C_out, K, C_in / groups)`` weights,
round-half-to-even, float32 / int32 defaults, no silent promotion to float64).

Nothing in the product, ``oracle/``, ``bench.py`` or the GPU tests imports this file.
"""
from __future__ import annotations

import builtins as _b
import math
import sys
import types
from typing import List, Optional

import numpy as np

# ----------------------------------------------------------------------------------------------- dtypes
float32, float64, float16 = np.float32, np.float64, np.float16
bfloat16 = np.float32  # no bfloat16 in numpy: fixtures are generated from float32 checkpoints
int32, int64, int16, int8, uint8, uint32 = np.int32, np.int64, np.int16, np.int8, np.uint8, np.uint32
bool_ = np.bool_
complex64 = np.complex64
pi = math.pi
inf = math.inf
nan = math.nan
newaxis = None
Dtype = type  # ``mx.Dtype`` only appears in annotations

_DOWN = {np.dtype(np.float64): np.float32, np.dtype(np.complex128): np.complex64, np.dtype(np.int64): np.int32}


class _At:
    def __init__(self, arr, idx=None):
        self.arr, self.idx = arr, idx

    def __getitem__(self, idx):
        return _At(self.arr, idx)

    def add(self, values):
        out = np.array(self.arr, copy=True)
        np.add.at(out, self.idx, np.asarray(values, dtype=out.dtype))
        return out.view(array)


class array(np.ndarray):
    """``mx.array``: an ndarray whose results never widen silently (MLX keeps float32 / int32 / complex64 unless float64 is asked for)."""

    def __new__(cls, val, dtype=None):
        if dtype is None:
            a = np.asarray(val)
            if a.dtype in _DOWN and not (isinstance(val, np.ndarray) and val.dtype == np.float64 and getattr(val, "_wide", False)):
                a = a.astype(_DOWN[a.dtype])
        else:
            a = np.asarray(val).astype(dtype)
        out = np.array(a, copy=True).view(cls)
        out._wide = dtype in (np.float64,)
        return out

    def __array_finalize__(self, obj):
        self._wide = getattr(obj, "_wide", False)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kw):
        wide = _b.any(isinstance(i, np.ndarray) and i.dtype == np.float64 and getattr(i, "_wide", False) for i in inputs)
        raw = [np.asarray(i) if isinstance(i, array) else i for i in inputs]
        if _b.any(isinstance(i, float) for i in raw):
            # MLX: a python float is a WEAK scalar -- against an integer array the result is float32 and the scalar is rounded to float32 first
            # (numpy would compute in float64 with the unrounded double: visible in interpolate's ``arange(size) * (W / size)``)
            raw = [np.float32(i) if isinstance(i, float) else (i.astype(np.float32) if isinstance(i, np.ndarray) and i.dtype.kind in "iub" else i)
                   for i in raw]
        if out is not None:
            kw["out"] = tuple(np.asarray(o) if isinstance(o, array) else o for o in out)
        res = getattr(ufunc, method)(*raw, **kw)
        if res is NotImplemented:
            return res

        def fix(r):
            if isinstance(r, np.ndarray) or isinstance(r, np.generic):
                r = np.asarray(r)
                if not wide and r.dtype in _DOWN:
                    r = r.astype(_DOWN[r.dtype])
                r = r.view(array)
                r._wide = wide
            return r

        return tuple(fix(r) for r in res) if isinstance(res, tuple) else fix(res)

    # MLX spellings
    def astype(self, dtype, stream=None):  # noqa: D401
        out = np.asarray(self).astype(dtype).view(array)
        out._wide = dtype in (np.float64,)
        return out

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        return np.transpose(np.asarray(self), axes if axes else None).view(array)

    def square(self):
        return self * self

    def mean(self, axis=None, keepdims=False, **kw):   # MLX takes a LIST of axes too (descript/nn/quantize.py:29: ``.mean([1, 2])``)
        ax = tuple(axis) if isinstance(axis, list) else axis
        return _wrap(np.asarray(np.mean(np.asarray(self), axis=ax, keepdims=keepdims, **kw)))

    def logsumexp(self, axis=None, keepdims=False):
        return logsumexp(self, axis=axis, keepdims=keepdims)

    def log(self):
        return log(self)

    def log10(self):
        return log10(self)

    def exp(self):
        return exp(self)

    def sqrt(self):
        return sqrt(self)

    def sin(self):
        return sin(self)

    def cos(self):
        return cos(self)

    def split(self, indices_or_sections, axis=0):
        return split(self, indices_or_sections, axis=axis)

    def moveaxis(self, source, destination):
        return np.moveaxis(np.asarray(self), source, destination).view(array)

    def flatten(self, start_axis=0, end_axis=-1):
        a = np.asarray(self)
        nd = a.ndim
        s0, e0 = start_axis % _b.max(nd, 1), end_axis % _b.max(nd, 1)
        return a.reshape(a.shape[:s0] + (-1,) + a.shape[e0 + 1:]).view(array)

    def abs(self):
        return np.abs(self)

    @property
    def at(self):
        return _At(self)

    def item(self):
        return np.asarray(self).item()

    def __hash__(self):
        return id(self)


def _wrap(x):
    if isinstance(x, array):
        return x
    if isinstance(x, (np.ndarray, np.generic)):
        a = np.asarray(x)
        if a.dtype in _DOWN:
            a = a.astype(_DOWN[a.dtype])
        return a.view(array)
    return x


def _f(fn):
    def g(*a, stream=None, **k):
        return _wrap(fn(*a, **k))
    g.__name__ = getattr(fn, "__name__", "f")
    return g


def _as(x):
    """python scalars become float32 / int32 scalars like MLX's weak-typed scalars against float32 arrays."""
    if isinstance(x, (float,)):
        return np.float32(x)
    return x


# ----------------------------------------------------------------------------------------------- mlx.core
def zeros(shape, dtype=np.float32, stream=None):
    return np.zeros(shape if not isinstance(shape, array) else tuple(shape), dtype=dtype).view(array)


def ones(shape, dtype=np.float32, stream=None):
    return np.ones(shape, dtype=dtype).view(array)


def full(shape, vals, dtype=None, stream=None):
    return array(np.full(shape, vals), dtype)


def zeros_like(a, stream=None):
    return np.zeros_like(np.asarray(a)).view(array)


def ones_like(a, stream=None):
    return np.ones_like(np.asarray(a)).view(array)


def arange(*args, dtype=None, stream=None):
    a = np.arange(*[np.asarray(v).item() if isinstance(v, np.ndarray) else v for v in args])
    if dtype is None:
        dtype = np.int32 if a.dtype.kind in "iu" else np.float32
    return a.astype(dtype).view(array)


def linspace(start, stop, num=50, dtype=np.float32, stream=None):
    return np.linspace(start, stop, int(num)).astype(dtype).view(array)


def concatenate(arrays, axis=0, stream=None):
    return _wrap(np.concatenate([np.asarray(a) for a in arrays], axis=axis))


def stack(arrays, axis=0, stream=None):
    return _wrap(np.stack([np.asarray(a) for a in arrays], axis=axis))


def expand_dims(a, axis, stream=None):
    return _wrap(np.expand_dims(np.asarray(a), axis))


def squeeze(a, axis=None, stream=None):
    return _wrap(np.squeeze(np.asarray(a), axis=axis))


def reshape(a, shape, stream=None):
    return _wrap(np.reshape(np.asarray(a), shape))


def transpose(a, axes=None, stream=None):
    return _wrap(np.transpose(np.asarray(a), axes))


def swapaxes(a, a1, a2, stream=None):
    return _wrap(np.swapaxes(np.asarray(a), a1, a2))


def broadcast_to(a, shape, stream=None):
    return _wrap(np.broadcast_to(np.asarray(a), shape))


def tile(a, reps, stream=None):
    return _wrap(np.tile(np.asarray(a), reps))


def repeat(a, repeats, axis=None, stream=None):
    return _wrap(np.repeat(np.asarray(a), int(np.asarray(repeats).item()) if np.ndim(repeats) == 0 else np.asarray(repeats), axis=axis))


def roll(a, shift, axis=None, stream=None):
    return _wrap(np.roll(np.asarray(a), shift, axis=axis))


def split(a, indices_or_sections, axis=0, stream=None):
    return [_wrap(p) for p in np.split(np.asarray(a), indices_or_sections, axis=axis)]


def pad(a, pad_width, mode="constant", constant_values=0, stream=None):
    a = np.asarray(a)
    if isinstance(pad_width, int):
        pad_width = [(pad_width, pad_width)] * a.ndim
    elif isinstance(pad_width, tuple) and len(pad_width) == 2 and _b.all(isinstance(p, (int, np.integer)) for p in pad_width):
        pad_width = [tuple(pad_width)] * a.ndim
    if mode == "constant":
        return _wrap(np.pad(a, pad_width, mode="constant", constant_values=constant_values))
    return _wrap(np.pad(a, pad_width, mode=mode))


def where(c, x, y, stream=None):
    x, y = _as(x), _as(y)
    r = np.where(np.asarray(c), np.asarray(x), np.asarray(y))
    if not _b.any(isinstance(v, np.ndarray) and v.dtype == np.float64 for v in (x, y)) and r.dtype in _DOWN:
        r = r.astype(_DOWN[r.dtype])
    return _wrap(r)


def _un(fn):
    def g(a, stream=None):
        a = np.asarray(_as(a))
        if a.dtype.kind in "iub":
            a = a.astype(np.float32)
        return _wrap(fn(a))
    return g


sin, cos, exp, log, sqrt, tanh, floor, ceil, arctan = (_un(f) for f in (np.sin, np.cos, np.exp, np.log, np.sqrt, np.tanh, np.floor, np.ceil, np.arctan))
abs = _f(np.abs)  # noqa: A001
real, imag = _f(np.real), _f(np.imag)


def arctan2(a, b, stream=None):
    return _wrap(np.arctan2(np.asarray(a), np.asarray(b)))


def sigmoid(a, stream=None):
    a = np.asarray(a)
    return _wrap((1.0 / (1.0 + np.exp(-a))).astype(a.dtype))


def rsqrt(a, stream=None):
    return _wrap(1.0 / np.sqrt(np.asarray(a)))


def square(a, stream=None):
    return _wrap(np.square(np.asarray(a)))


def power(a, b, stream=None):
    return _wrap(np.power(np.asarray(a), _as(b)))


def maximum(a, b, stream=None):
    return _wrap(np.maximum(np.asarray(_as(a)), np.asarray(_as(b))).astype(np.result_type(np.asarray(_as(a)).dtype, np.asarray(_as(b)).dtype)))


def minimum(a, b, stream=None):
    return _wrap(np.minimum(np.asarray(_as(a)), np.asarray(_as(b))).astype(np.result_type(np.asarray(_as(a)).dtype, np.asarray(_as(b)).dtype)))


def clip(a, a_min=None, a_max=None, stream=None):
    a = np.asarray(a)
    return _wrap(np.clip(a, None if a_min is None else np.asarray(_as(a_min)).astype(a.dtype) if np.ndim(a_min) == 0 else np.asarray(a_min),
                         None if a_max is None else np.asarray(_as(a_max)).astype(a.dtype) if np.ndim(a_max) == 0 else np.asarray(a_max)))


def round(a, decimals=0, stream=None):  # noqa: A001  half to even, like MLX
    return _wrap(np.round(np.asarray(a), decimals))


def nan_to_num(a, nan=0.0, posinf=None, neginf=None, stream=None):
    return _wrap(np.nan_to_num(np.asarray(a), nan=nan, posinf=posinf, neginf=neginf))


def matmul(a, b, stream=None):
    return _wrap(np.matmul(np.asarray(a), np.asarray(b)))


def addmm(c, a, b, alpha=1.0, beta=1.0, stream=None):
    return _wrap(np.float32(alpha) * np.matmul(np.asarray(a), np.asarray(b)) + np.float32(beta) * np.asarray(c))


def _red(fn):
    def g(a, axis=None, keepdims=False, stream=None):
        return _wrap(fn(np.asarray(a), axis=tuple(axis) if isinstance(axis, list) else axis, keepdims=keepdims))
    return g


sum, mean, max, min, prod = (_red(f) for f in (np.sum, np.mean, np.max, np.min, np.prod))  # noqa: A001


def var(a, axis=None, keepdims=False, ddof=0, stream=None):
    return _wrap(np.var(np.asarray(a), axis=tuple(axis) if isinstance(axis, list) else axis, keepdims=keepdims, ddof=ddof))


def std(a, axis=None, keepdims=False, ddof=0, stream=None):
    return _wrap(np.std(np.asarray(a), axis=tuple(axis) if isinstance(axis, list) else axis, keepdims=keepdims, ddof=ddof))


def cumsum(a, axis=None, reverse=False, inclusive=True, stream=None):
    assert not reverse and inclusive
    return _wrap(np.cumsum(np.asarray(a), axis=axis, dtype=np.asarray(a).dtype))


def softmax(a, axis=-1, precise=False, stream=None):
    a = np.asarray(a)
    e = np.exp(a - a.max(axis=axis, keepdims=True))
    return _wrap((e / e.sum(axis=axis, keepdims=True)).astype(a.dtype))


def as_strided(a, shape=None, strides=None, offset=0, stream=None):
    a = np.ascontiguousarray(np.asarray(a)).reshape(-1)[offset:]
    item = a.dtype.itemsize
    return _wrap(np.array(np.lib.stride_tricks.as_strided(a, shape=tuple(shape), strides=tuple(int(s) * item for s in strides)), copy=True))


def _t():
    import torch

    return torch


def conv1d(x, w, stride=1, padding=0, dilation=1, groups=1, stream=None):
    """x [N, L, C_in], w [C_out, K, C_in / groups] -> [N, L_out, C_out] (cross-correlation, like ``mx.conv1d``)."""
    torch = _t()
    xt = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).permute(0, 2, 1)
    wt = torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32))).permute(0, 2, 1)
    y = torch.nn.functional.conv1d(xt, wt, None, stride=int(stride), padding=int(padding), dilation=int(dilation), groups=int(groups))
    return _wrap(y.permute(0, 2, 1).contiguous().numpy())


def conv_transpose1d(x, w, stride=1, padding=0, dilation=1, output_padding=0, groups=1, stream=None):
    """x [N, L, C_in], w [C_out, K, C_in / groups] -> [N, (L - 1) * stride - 2 * padding + dilation * (K - 1) + 1 + output_padding, C_out]:
    out[n, t * stride + k * dilation - padding, co] += x[n, t, ci] * w[co, k, ci]  (``mx.conv_transpose1d``)."""
    torch = _t()
    xt = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).permute(0, 2, 1)
    wn = torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32)))
    g = int(groups)
    co, k, cig = wn.shape
    # torch wants (C_in, C_out / groups, K): group j owns input channels [j * cig, (j + 1) * cig) and output channels [j * co / g, ...)
    wt = wn.reshape(g, co // g, k, cig).permute(0, 3, 1, 2).reshape(g * cig, co // g, k)
    y = torch.nn.functional.conv_transpose1d(xt, wt, None, stride=int(stride), padding=int(padding), output_padding=int(output_padding),
                                             groups=g, dilation=int(dilation))
    return _wrap(y.permute(0, 2, 1).contiguous().numpy())


def logsumexp(a, axis=None, keepdims=False, stream=None):
    a = np.asarray(a)
    m = a.max(axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0).astype(a.dtype)
    r = np.log(np.exp(a - m).sum(axis=axis, keepdims=True)) + m
    return _wrap((r if keepdims else np.squeeze(r, axis=axis)).astype(a.dtype))


def argmax(a, axis=None, keepdims=False, stream=None):
    r = np.argmax(np.asarray(a), axis=axis, keepdims=keepdims)
    return _wrap(np.asarray(r).astype(np.uint32))


def all(a, axis=None, keepdims=False, stream=None):  # noqa: A001
    return _wrap(np.asarray(np.all(np.asarray(a), axis=axis, keepdims=keepdims)))


def any(a, axis=None, keepdims=False, stream=None):  # noqa: A001
    return _wrap(np.asarray(np.any(np.asarray(a), axis=axis, keepdims=keepdims)))


def moveaxis(a, source, destination, stream=None):
    return _wrap(np.moveaxis(np.asarray(a), source, destination))


def einsum(subscripts, *operands, stream=None):
    return _wrap(np.einsum(subscripts, *[np.asarray(o) for o in operands]))


def reciprocal(a, stream=None):
    return _wrap(1.0 / np.asarray(a))


def log10(a, stream=None):
    return _wrap(np.log10(np.asarray(a)))


def triu(a, k=0, stream=None):
    return _wrap(np.triu(np.asarray(a), k))


def tril(a, k=0, stream=None):
    return _wrap(np.tril(np.asarray(a), k))


def take(a, indices, axis=None, stream=None):
    return _wrap(np.take(np.asarray(a), np.asarray(indices), axis=axis))


def take_along_axis(a, indices, axis=None, stream=None):
    return _wrap(np.take_along_axis(np.asarray(a), np.asarray(indices).astype(np.int64), axis=axis))


def contiguous(a, stream=None):
    return _wrap(np.ascontiguousarray(np.asarray(a)))


def concat(arrays, axis=0, stream=None):
    return concatenate(arrays, axis=axis)


def argsort(a, axis=-1, stream=None):
    return _wrap(np.argsort(np.asarray(a), axis=axis, kind="stable").astype(np.uint32))


def sort(a, axis=-1, stream=None):
    return _wrap(np.sort(np.asarray(a), axis=axis, kind="stable"))


def argpartition(a, kth, axis=-1, stream=None):
    return _wrap(np.argpartition(np.asarray(a), kth, axis=axis).astype(np.uint32))


def put_along_axis(a, indices, values, axis=None, stream=None):
    out = np.array(np.asarray(a), copy=True)
    np.put_along_axis(out, np.asarray(indices).astype(np.int64), np.asarray(values), axis=axis)
    return _wrap(out)


class _Fast:
    """``mx.fast``: the fused ops, as their definitions (float32 accumulation, output in the input dtype)."""

    @staticmethod
    def scaled_dot_product_attention(q, k, v, *, scale, mask=None, sinks=None, stream=None):
        assert sinks is None
        q, k, v = (np.asarray(t, dtype=np.float32) for t in (q, k, v))
        B, Hq, Tq, D = q.shape
        Hk = k.shape[1]
        if Hk != Hq:  # grouped-query attention: head h reads kv head h // (Hq / Hk)
            k = np.repeat(k, Hq // Hk, axis=1)
            v = np.repeat(v, Hq // Hk, axis=1)
        sc = np.matmul(q * np.float32(scale), np.swapaxes(k, -1, -2))
        Tk = k.shape[2]
        if isinstance(mask, str):
            assert mask == "causal"
            qi = np.arange(Tk - Tq, Tk)[:, None]
            sc = np.where(qi >= np.arange(Tk)[None, :], sc, -np.inf)
        elif mask is not None:
            m = np.asarray(mask)
            sc = np.where(m, sc, -np.inf) if m.dtype == np.bool_ else sc + m.astype(np.float32)
        e = np.exp(sc - sc.max(axis=-1, keepdims=True))
        w = e / e.sum(axis=-1, keepdims=True)
        return _wrap(np.matmul(w, v).astype(np.float32))

    @staticmethod
    def rms_norm(x, weight, eps, stream=None):
        x = np.asarray(x, dtype=np.float32)
        y = x * (1.0 / np.sqrt((x * x).mean(axis=-1, keepdims=True) + np.float32(eps)))
        return _wrap(y if weight is None else y * np.asarray(weight, dtype=np.float32))

    @staticmethod
    def layer_norm(x, weight, bias, eps, stream=None):
        x = np.asarray(x, dtype=np.float32)
        y = (x - x.mean(axis=-1, keepdims=True)) * (1.0 / np.sqrt(x.var(axis=-1, keepdims=True) + np.float32(eps)))
        if weight is not None:
            y = y * np.asarray(weight)
        if bias is not None:
            y = y + np.asarray(bias)
        return _wrap(y.astype(np.float32))

    @staticmethod
    def rope(x, dims, *, traditional, base, scale, offset, freqs=None, stream=None):
        """x [..., T, D]: position t + offset rotates the first ``dims`` features; angle = pos * scale * base^(-2i / dims), or pos * scale / freqs[i].
        traditional: pairs (2i, 2i + 1); otherwise (i, i + dims / 2)."""
        x = np.asarray(x, dtype=np.float32)
        T = x.shape[-2]
        half = dims // 2
        if freqs is None:
            inv = np.float32(base) ** (-np.arange(0, half, dtype=np.float32) * np.float32(2.0 / dims))
        else:
            inv = (1.0 / np.asarray(freqs, dtype=np.float32)).astype(np.float32)
        off = np.asarray(offset)
        pos = (np.arange(T, dtype=np.float32) + np.float32(off.item() if off.ndim == 0 else 0)) * np.float32(scale)
        if off.ndim > 0:  # per-sequence offsets [B]
            pos = (np.arange(T, dtype=np.float32)[None, :] + off.astype(np.float32)[:, None]) * np.float32(scale)
            ang = pos[..., None] * inv
            ang = ang.reshape(ang.shape[0], *([1] * (x.ndim - 3)), T, half)
        else:
            ang = pos[:, None] * inv[None, :]
        c, s_ = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        out = np.array(x, copy=True)
        if traditional:
            a, b = x[..., 0:dims:2], x[..., 1:dims:2]
            out[..., 0:dims:2] = a * c - b * s_
            out[..., 1:dims:2] = a * s_ + b * c
        else:
            a, b = x[..., :half], x[..., half:dims]
            out[..., :half] = a * c - b * s_
            out[..., half:dims] = a * s_ + b * c
        return _wrap(out)


def _metal_kernel(name, input_names, output_names, header="", source="", **kw):
    """``mx.fast.metal_kernel``: custom Metal kernels cannot run here; the ONE kernel on the path (codec/models/encodec/encodec.py:89-123, ``lstm``) is
    restated from its Metal source: gate chunks i | f | g | o of h_in + x[:, t], sigmoid(x) = 1 / (1 + exp(-|x|)) mirrored for x < 0, precise tanh,
    cell = f * cell + i * g, hidden = o * tanh(cell).  The source's thread indexing (elem = b * 4H + y) is only consistent for one sequence per
    launch; that case is what is restated (and asserted)."""
    if name != "lstm":
        raise NotImplementedError(f"metal kernel {name!r} has no stand-in")

    def sig(x):
        y = 1.0 / (1.0 + np.exp(-np.abs(x)))
        return np.where(x < 0, 1.0 - y, y)

    def call(inputs, output_shapes, output_dtypes, grid, threadgroup, **k2):
        x, h_in, cell, hidden_size, time_step, num_time_steps = inputs
        x, h_in, cell = (np.asarray(t, dtype=np.float32) for t in (x, h_in, cell))
        assert x.shape[0] == 1 and x.shape[1] == int(num_time_steps), "the Metal kernel's indexing is only consistent for batch 1"
        H = int(hidden_size)
        g = h_in + x[:, int(time_step), :]
        i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sig(g[:, 3 * H:])
        c = (f * cell + i * gg).astype(np.float32)
        h = (o * np.tanh(c)).astype(np.float32)
        return _wrap(h), _wrap(c)

    return call


_Fast.metal_kernel = staticmethod(_metal_kernel)
fast = _Fast()


class _Distributed:
    class Group:  # noqa: D106
        pass

    @staticmethod
    def init(*a, **k):
        return None


distributed = _Distributed()


def eval(*a, **k):  # noqa: A001
    return None


def async_eval(*a, **k):
    return None


def compile(fn=None, *a, **k):  # noqa: A001
    return fn


def clear_cache():
    return None


def get_peak_memory():
    return 0


class _Stream:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def stream(s):
    return _Stream()


Stream = _Stream


cpu = gpu = object()


def new_stream(device=None):
    return _Stream()


def set_default_stream(s):
    return None


def dropout(x, p=0.5, *a, **k):
    raise RuntimeError("mx.dropout is not an MLX function; the reference never reaches this line in inference")


class _Random:
    """``mx.random``: seeded draws, every draw logged (kind, shape, values) so that the oracle can be fed the very same numbers."""

    state: list = []  # ``mx.random.state`` (only passed to mx.compile by the reference)

    def __init__(self):
        self.rng = np.random.default_rng(0)
        self.log: List[tuple] = []

    def seed(self, s):
        self.rng = np.random.default_rng(s)
        self.log = []

    def uniform(self, low=0.0, high=1.0, shape=(), dtype=np.float32, key=None, stream=None):
        v = (self.rng.uniform(size=tuple(shape)) * (np.asarray(high) - np.asarray(low)) + np.asarray(low)).astype(dtype)
        self.log.append(("uniform", tuple(shape), v))
        return v.view(array)

    def normal(self, shape=(), dtype=np.float32, loc=0.0, scale=1.0, key=None, stream=None):
        v = (self.rng.standard_normal(tuple(shape)) * scale + loc).astype(dtype)
        self.log.append(("normal", tuple(shape), v))
        return v.view(array)

    def randint(self, low, high, shape=(), dtype=np.int32, key=None, stream=None):
        return self.rng.integers(low, high, size=tuple(shape)).astype(dtype).view(array)


random = _Random()


class _FFT:
    @staticmethod
    def rfft(a, n=None, axis=-1, stream=None):
        return _wrap(np.fft.rfft(np.asarray(a, dtype=np.float64 if getattr(a, "_wide", False) else np.float32), n=n, axis=axis).astype(np.complex64))

    @staticmethod
    def irfft(a, n=None, axis=-1, stream=None):
        return _wrap(np.fft.irfft(np.asarray(a), n=n, axis=axis).astype(np.float32))


fft = _FFT()


# ----------------------------------------------------------------------------------------------- mlx.nn
class Module:
    """Attribute-based stand-in for ``mlx.nn.Module``: parameters are ``array`` attributes, children are Module attributes or lists of them."""

    training = True  # class default like MLX; ``eval()`` / ``train()`` set it on the whole tree

    def __init__(self):
        pass

    def __call__(self, *a, **k):
        raise NotImplementedError

    # MLX modules are dicts of their parameters / children
    def __contains__(self, key):
        return key in self.__dict__

    def __getitem__(self, key):
        return self.__dict__[key]

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def train(self, mode=True):
        for m in self.modules():
            m.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _children(self):
        for k, v in self.__dict__.items():
            if isinstance(v, Module):
                yield k, v
            elif isinstance(v, (list, tuple)):
                for i, e in enumerate(v):
                    if isinstance(e, Module):
                        yield f"{k}.{i}", e
                    elif isinstance(e, (list, tuple)):
                        for j, ee in enumerate(e):
                            if isinstance(ee, Module):
                                yield f"{k}.{i}.{j}", ee

    def named_modules(self, prefix=""):
        out = [(prefix, self)]
        for k, c in self._children():
            out.extend(c.named_modules(f"{prefix}.{k}" if prefix else k))
        return out

    def modules(self):
        return [m for _, m in self.named_modules()]

    def parameter_names(self, prefix=""):
        names = []
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            if isinstance(v, array):
                names.append(f"{prefix}{k}")
            elif isinstance(v, (list, tuple)) and v and _b.all(isinstance(e, array) for e in v):
                names.extend(f"{prefix}{k}.{i}" for i in range(len(v)))
        for k, c in self._children():
            names.extend(c.parameter_names(f"{prefix}{k}."))
        return names

    def load_weights(self, weights, strict=True):
        """Assigns by dotted path; returns nothing like MLX, but records ``_load_report`` = (missing, unexpected, shape_mismatches)."""
        have = set(self.parameter_names())
        given = dict(weights)
        mism = []
        for name, val in given.items():
            obj = self
            parts = name.split(".")
            try:
                for p in parts[:-1]:
                    obj = obj[int(p)] if isinstance(obj, (list, tuple)) else getattr(obj, p)
                last = parts[-1]
                new = array(np.asarray(val, dtype=np.float32))
                if isinstance(obj, list):
                    old = obj[int(last)]
                    obj[int(last)] = new
                else:
                    old = getattr(obj, last, None)
                    setattr(obj, last, new)
                if isinstance(old, np.ndarray) and old.shape != new.shape:
                    mism.append((name, old.shape, new.shape))
            except (AttributeError, IndexError, ValueError):
                pass
        self._load_report = (sorted(have - set(given)), sorted(set(given) - have), mism)
        return self


class Linear(Module):
    def __init__(self, input_dims, output_dims, bias=True):
        super().__init__()
        s = math.sqrt(1.0 / input_dims)
        self.weight = random.uniform(-s, s, (output_dims, input_dims))
        if bias:
            self.bias = random.uniform(-s, s, (output_dims,))

    def __call__(self, x):
        y = matmul(x, self.weight.T)
        return y + self.bias if "bias" in self.__dict__ else y


class Embedding(Module):
    def __init__(self, num_embeddings, dims):
        super().__init__()
        self.weight = random.normal((num_embeddings, dims), scale=math.sqrt(1.0 / dims))

    def __call__(self, x):
        return _wrap(np.asarray(self.weight)[np.asarray(x)])

    def as_linear(self, x):
        return matmul(x, self.weight.T)


class LayerNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=True, bias=True):
        super().__init__()
        self.eps, self.dims = eps, dims
        if affine:
            self.weight = ones((dims,))
            if bias:
                self.bias = zeros((dims,))

    def __call__(self, x):
        x = np.asarray(x)
        mu = x.mean(axis=-1, keepdims=True)
        v = x.var(axis=-1, keepdims=True)
        y = (x - mu) * (1.0 / np.sqrt(v + np.float32(self.eps)))
        if "weight" in self.__dict__:
            y = y * np.asarray(self.weight)
            if "bias" in self.__dict__:
                y = y + np.asarray(self.bias)
        return _wrap(y.astype(x.dtype))


class InstanceNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=False):
        super().__init__()
        self.eps = eps

    def __call__(self, x):  # channels last: normalise over the spatial axes
        x = np.asarray(x)
        ax = tuple(range(1, x.ndim - 1))
        mu = x.mean(axis=ax, keepdims=True)
        v = x.var(axis=ax, keepdims=True)
        return _wrap(((x - mu) * (1.0 / np.sqrt(v + np.float32(self.eps)))).astype(x.dtype))


class Conv1d(Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        s = math.sqrt(1.0 / (in_channels * kernel_size))
        self.weight = random.uniform(-s, s, (out_channels, kernel_size, in_channels // groups))
        if bias:
            self.bias = zeros((out_channels,))
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups

    def __call__(self, x):
        y = conv1d(x, self.weight, self.stride, self.padding, self.dilation, self.groups)
        return y + self.bias if "bias" in self.__dict__ else y


class Dropout(Module):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def __call__(self, x):
        return x  # inference


class Identity(Module):
    def __init__(self, *a, **k):
        super().__init__()

    def __call__(self, x, *a, **k):
        return x


class Upsample(Module):
    """Channels-last nearest up-sampling by an integer factor along every spatial axis (the only mode the reference's models use)."""

    def __init__(self, scale_factor, mode="nearest", align_corners=False):
        super().__init__()
        assert mode == "nearest"
        # a plain number (KittenTTS passes an mx.array product): not a parameter of the module
        self.scale_factor = tuple(scale_factor) if isinstance(scale_factor, (tuple, list)) else float(np.asarray(scale_factor).item())

    def __call__(self, x):
        s = int(self.scale_factor) if not isinstance(self.scale_factor, tuple) else None
        x = np.asarray(x)
        for ax in range(1, x.ndim - 1):
            x = np.repeat(x, s if s is not None else int(self.scale_factor[ax - 1]), axis=ax)
        return _wrap(x)


class RMSNorm(Module):
    def __init__(self, dims, eps=1e-5):
        super().__init__()
        self.weight = ones((dims,))
        self.eps = eps

    def __call__(self, x):
        return fast.rms_norm(x, self.weight, self.eps)


class RoPE(Module):
    def __init__(self, dims, traditional=False, base=10000, scale=1.0):
        super().__init__()
        self.dims, self.traditional, self.base, self.scale = dims, traditional, base, scale

    def __call__(self, x, offset=0):
        return fast.rope(x, self.dims, traditional=self.traditional, base=self.base, scale=self.scale, offset=offset)


class ConvTranspose1d(Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, output_padding=0, bias=True):
        super().__init__()
        s = math.sqrt(1.0 / (in_channels * kernel_size))
        self.weight = random.uniform(-s, s, (out_channels, kernel_size, in_channels))
        if bias:
            self.bias = zeros((out_channels,))
        self.stride, self.padding, self.dilation, self.output_padding = stride, padding, dilation, output_padding

    def __call__(self, x):
        y = conv_transpose1d(x, self.weight, self.stride, self.padding, self.dilation, self.output_padding)
        return y + self.bias if "bias" in self.__dict__ else y


def elu(x, alpha=1.0):
    x = np.asarray(x)
    return _wrap(np.where(x > 0, x, np.float32(alpha) * (np.exp(np.minimum(x, 0)) - 1)).astype(x.dtype))


class ELU(Module):
    def __init__(self, alpha=1.0):
        super().__init__()
        self._alpha = alpha

    def __call__(self, x):
        return elu(x, self._alpha)


def gelu_approx(x):
    torch = _t()
    return _wrap(torch.nn.functional.gelu(torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))), approximate="tanh").numpy())


class Sequential(Module):
    def __init__(self, *modules):
        super().__init__()
        self.layers = list(modules)

    def __call__(self, x):
        for m in self.layers:
            x = m(x)
        return x


class Tanh(Module):
    def __call__(self, x):
        return tanh(x)


def log_softmax(x, axis=-1):
    x = np.asarray(x)
    m = x.max(axis=axis, keepdims=True)
    return _wrap((x - m - np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).astype(x.dtype))


class MultiHeadAttention(Module):
    """Only the static helper the reference's Whisper uses (whisper.py:468)."""

    @staticmethod
    def create_additive_causal_mask(N, dtype=np.float32):
        idx = np.arange(N)
        # MLX: (indices[:, None] < indices[None]) * finfo(dtype).min
        return _wrap(((idx[:, None] < idx[None]).astype(np.float32) * np.finfo(np.float32).min).astype(dtype))


def leaky_relu(x, negative_slope=0.01):
    x = np.asarray(x)
    return _wrap(np.maximum(np.float32(negative_slope) * x, x))


class LeakyReLU(Module):
    def __init__(self, negative_slope=0.01):
        super().__init__()
        self.negative_slope = negative_slope

    def __call__(self, x):
        return leaky_relu(x, self.negative_slope)


def gelu(x):
    torch = _t()
    return _wrap(torch.nn.functional.gelu(torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))).numpy())


class GELU(Module):
    def __init__(self, approx="none"):
        super().__init__()
        assert approx in ("none", None)

    def __call__(self, x):
        return gelu(x)


def nn_tanh(x):
    return tanh(x)


def relu(x):
    return _wrap(np.maximum(np.asarray(x), 0))


def silu(x):
    x = np.asarray(x)
    return _wrap(x * (1.0 / (1.0 + np.exp(-x))))


# ----------------------------------------------------------------------------------------------- registration
def install():
    """Registers ``mlx``, ``mlx.core``, ``mlx.nn``, ``mlx.utils`` in ``sys.modules`` (refuses to shadow a real MLX)."""
    if "mlx" in sys.modules and not getattr(sys.modules["mlx"], "_IS_SHIM", False):
        raise RuntimeError("a real mlx is importable: use it instead of the stand-in")
    me = sys.modules[__name__]
    core = types.ModuleType("mlx.core")
    for k, v in vars(me).items():
        if not k.startswith("_") and k not in ("Module", "Linear", "Embedding", "LayerNorm", "InstanceNorm", "Conv1d", "Dropout", "Identity", "Upsample",
                                               "LeakyReLU", "GELU", "leaky_relu", "gelu", "nn_tanh", "relu", "silu", "install", "MultiHeadAttention", "RMSNorm", "RoPE", "ConvTranspose1d", "elu", "gelu_approx", "Sequential", "Tanh", "log_softmax", "ELU"):
            setattr(core, k, v)
    nn = types.ModuleType("mlx.nn")
    for k in ("Module", "Linear", "Embedding", "LayerNorm", "InstanceNorm", "Conv1d", "Dropout", "Identity", "Upsample", "LeakyReLU", "GELU", "leaky_relu",
              "gelu", "relu", "silu", "sigmoid", "MultiHeadAttention", "RMSNorm", "RoPE", "ConvTranspose1d", "elu", "gelu_approx", "Sequential", "Tanh", "log_softmax", "ELU"):
        setattr(nn, k, getattr(me, k))
    nn.tanh = nn_tanh
    losses = types.ModuleType("mlx.nn.losses")

    def mse_loss(predictions, targets, reduction="mean"):
        """``mlx.nn.losses.mse_loss``: ``square(predictions - targets)``, then ``none`` / ``mean`` / ``sum``."""
        loss = _wrap(np.square(np.asarray(predictions) - np.asarray(targets)))
        return loss if reduction == "none" else (_wrap(np.asarray(loss.mean())) if reduction == "mean" else _wrap(np.asarray(loss.sum())))

    losses.mse_loss = mse_loss
    nn.losses = losses
    sys.modules["mlx.nn.losses"] = losses
    utils = types.ModuleType("mlx.utils")
    utils.tree_flatten = lambda tree: []

    def tree_map(fn, tree, *rest):
        if isinstance(tree, (list, tuple)):
            return type(tree)(tree_map(fn, t, *[r[i] for r in rest]) for i, t in enumerate(tree))
        if isinstance(tree, dict):
            return {k: tree_map(fn, v, *[r[k] for r in rest]) for k, v in tree.items()}
        return fn(tree, *rest)

    utils.tree_map = tree_map
    utils.tree_unflatten = lambda pairs: dict(pairs)
    utils.tree_reduce = lambda fn, tree, init=None: init
    root = types.ModuleType("mlx")
    root._IS_SHIM = True
    root.core, root.nn, root.utils = core, nn, utils
    root.__path__ = []
    layers = types.ModuleType("mlx.nn.layers")
    dist_l = types.ModuleType("mlx.nn.layers.distributed")
    dist_l.shard_linear = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("tensor parallel sharding is not part of the fixtures"))
    layers.distributed = dist_l
    nn.layers = layers
    nn.__path__ = []
    layers.__path__ = []
    sys.modules.update({"mlx": root, "mlx.core": core, "mlx.nn": nn, "mlx.utils": utils, "mlx.nn.layers": layers, "mlx.nn.layers.distributed": dist_l})
    return core, nn
