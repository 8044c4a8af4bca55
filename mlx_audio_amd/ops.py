"""Thin torch-tensor front end of the C ABI (include/mi355audio.h).

PyTorch is plumbing only: it owns device memory (caching allocator) and streams; every compute op
below is a hand-written gfx950 kernel reached through ctypes.  All activations are float32,
channels-last ``[B, L, C]`` views whose last stride is 1 (slices of wider buffers are fine: that is
how channel concatenation is fused away).
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_LEAKY, ACT_SNAKE, ACT_GELU, ACT_ELU, ACT_SILU, ACT_GELU_TANH, ACT_TANH = 0, 1, 2, 3, 4, 5, 6, 7

# bench.py's roofline leg: when a list is installed here every conv_gemm launch is bracketed by
# events on the launch stream and (algorithmic flops, algorithmic bytes, start, end) is appended.
PROFILE = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def profile_finalize(entries):
    """PROFILE entries -> [(algorithmic flops, algorithmic bytes, start event, end event, (cin, cout, k, dil, rows))] after a synchronize.

    Algorithmic bytes of one launch = every operand once: fp32 input rows, fp32 output rows, the 16-bit weight image, plus one more fp32
    output-shaped read per residual operand and per ``accumulate`` (the launch must read what it adds to) -- the figure the PMC traffic is
    compared with."""
    torch.cuda.synchronize()
    out = []
    for rows, cin, cout, k, dil, extra_reads, e0, e1 in entries:
        rows = int(rows)
        flops = 2.0 * rows * cout * k * cin
        byts = 4.0 * rows * (cin + cout * (1 + extra_reads)) + 2.0 * cout * k * cin
        out.append((flops, byts, e0, e1, (cin, cout, k, dil, rows)))
    return out


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _nlc(t: torch.Tensor) -> Tuple[int, int, int, int, int]:
    """(B, L, C, bstride, ld) of a channels-last view."""
    assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)
    return t.shape[0], t.shape[1], t.shape[2], t.stride(0), t.stride(1)


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.Mi355Error("mlx_audio_amd needs a ROCm device (MI355X / gfx950); there is no CPU fallback")
    _lib.load()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------- weights
@dataclass
class PackedConv:
    """bf16 weights of one conv / linear in MFMA fragment order plus its fp32 bias (device)."""

    w: torch.Tensor  # int16 view of the packed bf16 data, on device
    bias: Optional[torch.Tensor]
    cout: int
    k: int
    cin: int
    f16: bool = False  # packed as fp16 for the single-pass fp16 MFMA mode (precision 3)
    mx: int = 0        # 1: MX image: fp16 tap slices + e4m3 tap-pair slices + E8M0 column scales (precision 5); 2: MX4 image, FP4 (e2m1) tap-pair slices (precision 6)


def mx_eligible(cout: int, k: int, cin: int) -> bool:
    """Shapes the wave-specialised kernel takes at precision 5 (``mi355_conv_ws4_eligible``): K = 3 (mod 4), more than 64 outputs, >= 64 inputs."""
    return k % 4 == 3 and cout > 64 and cin >= 64


def mx_pays(cout: int, k: int, cin: int) -> bool:
    """... and for which it is the faster arithmetic (profiles/r4_conv_prec_ab_b32_call3.txt): from 7 taps on.  The 3-tap convs are bound by HBM and
    by the producers' VALU work, where the bf16 split is the cheapest prologue (cvt_pk + shift / mask): they stay on the bf16 hi + lo pass."""
    return mx_eligible(cout, k, cin) and k >= 7


def pack_conv(w: torch.Tensor, bias: Optional[torch.Tensor], device, f16: bool = False, mx=False) -> PackedConv:
    """``w``: float32 CPU tensor ``[Cout, K, Cin]`` (MLX conv layout) or ``[Cout, Cin]`` (linear).  ``mx``: True / 1 = the image of precision 5 (e4m3 lo
    slices), 2 = the MX4 image of precision 6 (FP4 lo slices)."""
    if w.dim() == 2:
        w = w[:, None, :]
    w = w.detach().to(torch.float32).contiguous().cpu()
    cout, k, cin = w.shape
    lib = _lib.load()
    if mx:
        nb = lib.mi355_packed_conv_weight_mx_bytes(cout, k, cin)
        out8 = np.empty(nb, dtype=np.uint8)
        fn = "mi355_pack_conv_weight_mx4_host" if int(mx) == 2 else "mi355_pack_conv_weight_mx_host"
        rc = getattr(lib, fn)(w.numpy().ctypes.data, cout, k, cin, out8.ctypes.data)
        _lib.check(rc, fn)
        wd = torch.from_numpy(out8).to(device)
        bd = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)
        return PackedConv(wd, bd, cout, k, cin, True, 2 if int(mx) == 2 else 1)
    n = lib.mi355_packed_conv_weight_elems(cout, k, cin)
    out = np.empty(n, dtype=np.uint16)
    rc = lib.mi355_pack_conv_weight_host_dt(w.numpy().ctypes.data, cout, k, cin, 1 if f16 else 0, out.ctypes.data)
    _lib.check(rc, "mi355_pack_conv_weight_host_dt")
    wd = torch.from_numpy(out.view(np.int16)).to(device)
    bd = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)
    return PackedConv(wd, bd, cout, k, cin, f16)


def pack_conv_transpose(w_t: torch.Tensor, bias: Optional[torch.Tensor], stride: int, device, f16: bool = False) -> PackedConv:
    """Polyphase repack of a transposed conv.  ``w_t``: ``[Cout, K, Cin]`` as ``mx.conv_transpose1d``
    takes it (out[n] += x[t] * w_t[:, k, :] for n = t*stride + k - pad); K must be a multiple of
    stride.  Returns the equivalent stride-1 conv with K/stride taps and ``stride*Cout`` outputs:
    GEMM row u, column r*Cout+co  ->  out[u*stride + r - pad, co]."""
    return pack_conv(_polyphase_weight(w_t.to(torch.float32), stride), bias, device, f16)


def _polyphase_weight(w_t: torch.Tensor, stride: int) -> torch.Tensor:
    """[Cout, K, Cin] transposed-conv weight -> [stride*Cout, K/stride, Cin] stride-1 conv weight (host logic)."""
    cout, k, cin = w_t.shape
    assert k % stride == 0, "polyphase conv_transpose needs K % stride == 0"
    kp = k // stride
    w = torch.empty((stride * cout, kp, cin), dtype=w_t.dtype)
    for r in range(stride):
        for tp in range(kp):
            w[r * cout:(r + 1) * cout, tp, :] = w_t[:, r + (kp - 1 - tp) * stride, :]
    return w


@dataclass
class RowMajor16:
    """Row-major 16-bit weight image [N, K] for the decode-step GEMV (checkpoint dtype: bf16 or fp16)."""

    w: torch.Tensor  # int16 [N, K] on device (uint8 [N, K] for the fp8 image)
    bias: Optional[torch.Tensor]
    n: int
    k: int
    f16: bool
    scale: Optional[torch.Tensor] = None  # fp8 image only: per-row power-of-two dequantisation scale [N] fp32 (device)
    interleaved: bool = False             # recurrent weights of mi355_lstm_seq with their rows ordered by hidden unit (row 4 j + gate): the one-launch step

    @property
    def wdtype(self) -> int:
        return W_FP8 if self.scale is not None else (W_F16 if self.f16 else W_BF16)


def pack_rowmajor16(w: torch.Tensor, bias: Optional[torch.Tensor], device, f16: bool = False) -> RowMajor16:
    """``w``: float32 CPU tensor [N, K] (nn.Linear weight layout)."""
    w = w.detach().to(torch.float32).contiguous().cpu()
    n, k = w.shape
    assert k % 8 == 0, "gemv needs K % 8 == 0"
    lib = _lib.load()
    out = np.empty(n * k, dtype=np.uint16)
    rc = lib.mi355_pack_rowmajor16_host(w.numpy().ctypes.data, n * k, 1 if f16 else 0, out.ctypes.data)
    _lib.check(rc, "mi355_pack_rowmajor16_host")
    wd = torch.from_numpy(out.view(np.int16).reshape(n, k)).to(device)
    bd = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)
    return RowMajor16(wd, bd, n, k, f16)


W_BF16, W_F16, W_FP8 = 0, 1, 2  # MI355_W_* of the header


def quantize_rows_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """float32 [N, K] (CPU) -> (uint8 [N, K] OCP e4m3fn codes, float32 [N] power-of-two row scales) through the library's host packer
    (``mi355_pack_rowmajor_fp8_host``): w ~= decode(code) * scale, and decode(code) * scale is exactly representable in bf16."""
    w = w.detach().to(torch.float32).contiguous().cpu()
    n, k = w.shape
    assert k % 16 == 0, "fp8 GEMV images need K % 16 == 0"
    lib = _lib.load()
    codes = np.empty((n, k), dtype=np.uint8)
    scale = np.empty(n, dtype=np.float32)
    rc = lib.mi355_pack_rowmajor_fp8_host(w.numpy().ctypes.data, n, k, codes.ctypes.data, scale.ctypes.data)
    _lib.check(rc, "mi355_pack_rowmajor_fp8_host")
    return torch.from_numpy(codes), torch.from_numpy(scale)


def dequantize_rows_fp8(codes: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """The float32 weights an fp8 image stands for (exact): used to build the bf16 MFMA image of the SAME quantised Linear for prefill."""
    return codes.view(torch.float8_e4m3fn).to(torch.float32) * scale[:, None]


def pack_rowmajor_fp8(w: torch.Tensor, bias: Optional[torch.Tensor], device) -> Tuple[RowMajor16, torch.Tensor]:
    """fp8 GEMV image of an nn.Linear weight [N, K]; also returns the dequantised float32 weights (CPU) it represents."""
    codes, scale = quantize_rows_fp8(w)
    bd = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)
    rw = RowMajor16(codes.to(device), bd, codes.shape[0], codes.shape[1], False, scale.to(device))
    return rw, dequantize_rows_fp8(codes, scale)


def gemv(x: torch.Tensor, rw: RowMajor16, y: torch.Tensor, *, post_act: int = ACT_NONE, post_slope: float = 0.0,
         res: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None, out_scale: float = 1.0, glu: bool = False,
         use_bias: bool = True, norm: Optional[tuple] = None, y2: Optional[torch.Tensor] = None, rope: Optional[tuple] = None,
         x_ids: Optional[torch.Tensor] = None, x_id_offset: int = 0, w_policy: int = 0):
    """y[m, :] = epilogue(norm(x[m, :]) @ W^T) for 1..8 rows (any weight image) or 9..64 rows (16-bit images, K % 64 == 0); x / y / res are 2-D fp32 views with unit inner stride.
    ``norm`` = (mode, weight, bias, eps) with mode "layer" | "rms" fuses the input normalisation; ``y2``: columns >= y.shape[1] go there."""
    assert x.dim() == 2 and y.dim() == 2 and x.stride(1) == 1 and y.stride(1) == 1 and x.dtype == torch.float32 and y.dtype == torch.float32
    M = x.shape[0] if x_ids is None else 1   # x_ids int32 [1] (device): x is then a TABLE and the input row is x[x_ids[0] + x_id_offset]
    n_y = rw.n // 2 if glu else (rw.n if y2 is None else rw.n - y2.shape[1])
    assert x.shape[1] == rw.k and y.shape[0] == M and y.shape[1] == n_y, (x.shape, y.shape, rw.n, rw.k)
    kw = dict(x=_ptr(x), ldx=x.stride(0), M=M, K=rw.k, w=_ptr(rw.w), ldw=rw.w.stride(0), wdtype=rw.wdtype, wscale=_ptr(rw.scale), N=rw.n,
              bias=_ptr(rw.bias) if use_bias else None, post_act=post_act, post_slope=post_slope, colscale=_ptr(colscale),
              out_scale=out_scale, glu=int(glu), y=_ptr(y), ldy=y.stride(0), w_policy=int(w_policy))
    if res is not None:
        assert res.dim() == 2 and res.stride(1) == 1
        kw.update(res=_ptr(res), ldr=res.stride(0))
    if norm is not None:
        mode, nw, nb, eps = norm
        kw.update(norm={"layer": 1, "rms": 2}[mode], norm_weight=_ptr(nw), norm_bias=_ptr(nb), norm_eps=eps)
    if y2 is not None:
        assert y2.dim() == 2 and y2.stride(1) == 1 and y2.shape[0] == M and y2.dtype in KV_DTYPES
        kw.update(y2=_ptr(y2), ldy2=y2.stride(0), split=n_y, y2_dtype=KV_DTYPES[y2.dtype])
    if 5 <= M <= 8 and rw.k > 2048 and norm is None and not glu and rw.wdtype != 2:  # K split over workgroups for the long-K projections
        sws, scnt = gemv_split_workspace(x.device, rw.n, rw.k)
        kw.update(split_ws=_ptr(sws), split_cnt=_ptr(scnt))
    if x_ids is not None:
        assert x_ids.dtype == torch.int32 and x_ids.numel() == 1 and x_ids.is_cuda
        kw.update(x_ids=_ptr(x_ids), x_id_offset=int(x_id_offset))
    if rope is not None:  # (cos_row [dh / 2], sin_row [dh / 2], dh, cols): interleaved rotary pairs on the first ``cols`` output columns
        cos_row, sin_row, dh, cols = rope
        assert cos_row.is_contiguous() and sin_row.is_contiguous() and cos_row.numel() >= dh // 2
        kw.update(rope_cos=_ptr(cos_row), rope_sin=_ptr(sin_row), rope_dh=dh, rope_cols=cols)
    _lib.call_struct("mi355_gemv", "mi355_gemv_args", _stream(), **kw)
    return y


# --------------------------------------------------------------------------------------- decode steps for 9..64 sequences (rows_pipe.hip)
@dataclass
class Tiles16:
    """Tile image of a [N, K] Linear for ``mi355_rows_gemm``: [ceil(N / 16)][K / 64][4 groups][16 rows][16 elements] 16-bit elements."""

    w: torch.Tensor  # int16 (fp8 image: uint8), flat, on the device
    n: int
    k: int
    f16: bool
    scale: Optional[torch.Tensor] = None  # fp8 image only: per-row power-of-two dequantisation scale [N] fp32 (device)

    @property
    def wdtype(self) -> int:
        return W_FP8 if self.scale is not None else (W_F16 if self.f16 else W_BF16)


def tiles16_from_rowmajor(rm: RowMajor16) -> Tiles16:
    """The tile image of a row-major 16-bit image, permuted on the device (== ``mi355_pack_tiles16_host`` of the same weights, element for element)."""
    assert rm.k % 64 == 0, "tile images need K % 64 == 0"   # an fp8 image (uint8 codes + per-row scales) is permuted the same way, one byte per element
    nt = (rm.n + 15) // 16
    w = rm.w[:, : rm.k]
    if nt * 16 != rm.n:
        w = torch.cat([w, torch.zeros((nt * 16 - rm.n, rm.k), dtype=w.dtype, device=w.device)], 0)
    t = w.reshape(nt, 16, rm.k // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)
    return Tiles16(t, rm.n, rm.k, rm.f16, rm.scale)


def pack_tiles8_host(codes: torch.Tensor) -> np.ndarray:
    """``mi355_pack_tiles8_host`` on uint8 e4m3 codes [N, K] (CPU): uint8 [ceil(N / 16) * 16 * K]."""
    c = codes.detach().to(torch.uint8).contiguous().cpu()
    n, k = c.shape
    out = np.empty(((n + 15) // 16) * 16 * k, dtype=np.uint8)
    rc = _lib.load().mi355_pack_tiles8_host(c.numpy().ctypes.data, n, k, out.ctypes.data)
    _lib.check(rc, "mi355_pack_tiles8_host")
    return out


def pack_tiles16_host(w: torch.Tensor, f16: bool = False) -> np.ndarray:
    """``mi355_pack_tiles16_host`` on float32 CPU weights [N, K]: uint16 [ceil(N / 16) * 16 * K]."""
    w = w.detach().to(torch.float32).contiguous().cpu()
    n, k = w.shape
    out = np.empty(((n + 15) // 16) * 16 * k, dtype=np.uint16)
    rc = _lib.load().mi355_pack_tiles16_host(w.numpy().ctypes.data, n, k, 1 if f16 else 0, out.ctypes.data)
    _lib.check(rc, "mi355_pack_tiles16_host")
    return out


def rows_R(M: int) -> int:
    """Rows of the planes that carry ``M`` sequences: 16, 32 or 64."""
    assert 1 <= M <= 64
    return 16 if M <= 16 else (32 if M <= 32 else 64)


def rows_kgroups(n: int, k: int) -> int:
    return int(_lib.load().mi355_rows_kgroups(n, k))


def rows_planes(R: int, k: int, device) -> torch.Tensor:
    """Storage of the planes of R rows x k columns (hi + lo images: 4 bytes per element)."""
    return torch.zeros(2 * R * k, dtype=torch.int16, device=device)


def rows_gemm(planes: torch.Tensor, tl: Tiles16, part: Optional[torch.Tensor], M: int, R: int, kgroups: Optional[int] = None, *,
              glu_planes_out: Optional[torch.Tensor] = None, glu_bias: Optional[torch.Tensor] = None):
    """part[g, :M, :N] = partial products of the g-th range of k steps; ``part`` is fp32 [kgroups, rows >= M, ld >= N] (contiguous slabs).
    ``glu_planes_out`` (one K group): silu(gate + b) * (up + b) of the (gate, up) column pairs leaves as planes instead."""
    kg = rows_kgroups(tl.n, tl.k) if kgroups is None else kgroups
    if glu_planes_out is not None:
        assert kg == 1 and glu_planes_out.numel() * glu_planes_out.element_size() >= 4 * R * (tl.n // 2)
        _lib.call_struct("mi355_rows_gemm", "mi355_rows_gemm_args", _stream(), wt=_ptr(tl.w), wdtype=tl.wdtype, N=tl.n, K=tl.k, planes=_ptr(planes), M=M, R=R,
                         kgroups=1, glu_planes_out=_ptr(glu_planes_out), glu_bias=_ptr(glu_bias), wscale=_ptr(tl.scale))
        return 1
    assert part.dtype == torch.float32 and part.dim() == 3 and part.shape[0] >= kg and part.shape[1] >= M and part.shape[2] >= tl.n and part.stride(2) == 1
    assert planes.numel() * planes.element_size() >= 4 * R * tl.k
    _lib.call_struct("mi355_rows_gemm", "mi355_rows_gemm_args", _stream(), wt=_ptr(tl.w), wdtype=tl.wdtype, N=tl.n, K=tl.k, planes=_ptr(planes), M=M, R=R,
                     part=_ptr(part), ldp=part.stride(1), kg_stride=part.stride(0), kgroups=kg)
    return kg


def rows_finish(part: torch.Tensor, M: int, N: int, kgroups: int = 1, *, bias=None, post_act: int = ACT_NONE, post_slope: float = 0.0, colscale=None,
                res: Optional[torch.Tensor] = None, out_scale: float = 1.0, glu: bool = False, wscale: Optional[torch.Tensor] = None,
                y: Optional[torch.Tensor] = None, y2: Optional[torch.Tensor] = None, norm: Optional[tuple] = None, yn: Optional[torch.Tensor] = None,
                planes: Optional[torch.Tensor] = None, R: int = 0, f16: bool = False):
    """Row epilogue of ``rows_gemm`` (see ``mi355_rows_finish_args``).  ``part``: fp32 [kgroups, rows, ld] slabs, or a 2-D fp32 matrix (kgroups = 1:
    the converter rows -> planes).  ``y`` / ``y2`` / ``yn`` are 2-D views with unit inner stride; ``norm`` = (mode, weight, bias, eps)."""
    if part.dim() == 2:
        part = part.unsqueeze(0)
    assert part.dtype == torch.float32 and part.stride(2) == 1 and part.shape[0] >= kgroups
    kw = dict(part=_ptr(part), kgroups=kgroups, kg_stride=part.stride(0) if kgroups > 1 else 0, ldp=part.stride(1), M=M, N=N, bias=_ptr(bias),
              post_act=post_act, post_slope=post_slope, colscale=_ptr(colscale), out_scale=out_scale, glu=int(glu), wscale=_ptr(wscale))
    n_out = N // 2 if glu else N
    if res is not None:
        assert res.dim() == 2 and res.stride(1) == 1
        kw.update(res=_ptr(res), ldr=res.stride(0))
    if y2 is not None:
        assert y is not None and y2.dim() == 2 and y2.stride(1) == 1 and y2.dtype in KV_DTYPES
        kw.update(y2=_ptr(y2), ldy2=y2.stride(0), split=n_out - y2.shape[1], y2_dtype=KV_DTYPES[y2.dtype])
    if y is not None:
        assert y.dim() == 2 and y.stride(1) == 1 and y.dtype == torch.float32
        kw.update(y=_ptr(y), ldy=y.stride(0))
    if norm is not None:
        mode, nw, nb, eps = norm
        kw.update(norm={"layer": 1, "rms": 2}[mode], norm_weight=_ptr(nw), norm_bias=_ptr(nb), norm_eps=eps)
    if yn is not None:
        assert yn.dim() == 2 and yn.stride(1) == 1 and yn.dtype == torch.float32
        kw.update(yn=_ptr(yn), ldyn=yn.stride(0))
    if planes is not None:
        assert R in (16, 32, 64) and planes.numel() * planes.element_size() >= 4 * R * n_out
        kw.update(planes=_ptr(planes), R=R, planes_dtype=W_F16 if f16 else W_BF16)
    _lib.call_struct("mi355_rows_finish", "mi355_rows_finish_args", _stream(), **kw)


def pack_lstm_wh(wh_f: torch.Tensor, wh_b: torch.Tensor, device, f16: bool = False) -> torch.Tensor:
    """Recurrent weights of both directions in the persistent kernel's layout; ``f16``: IEEE half instead of bf16 values (float32 checkpoints; pass
    ``wh_f16=True`` to ``lstm_bidir``)."""
    H = wh_f.shape[1]
    lib = _lib.load()
    out = np.empty(2 * 4 * H * H, dtype=np.uint16)
    a = wh_f.detach().to(torch.float32).contiguous().cpu().numpy()
    b = wh_b.detach().to(torch.float32).contiguous().cpu().numpy()
    rc = lib.mi355_pack_lstm_wh16_host(a.ctypes.data, b.ctypes.data, H, int(bool(f16)), out.ctypes.data)
    _lib.check(rc, "mi355_pack_lstm_wh16_host")
    return torch.from_numpy(out.view(np.int16)).to(device)


def pack_lstm_wh_scaled(wh_f: torch.Tensor, wh_b: torch.Tensor, device):
    """bf16-valued recurrent weights as an IEEE-half image scaled by a power of two into half's normal range: (image, wh_scale) for
    ``lstm_bidir(wh_f16=True, wh_scale=...)``, or ``None`` when some value would not be held exactly (then use the bf16 image).  Exact because a bf16
    value has 8 significant bits and half holds 11 in its normal range; the kernel's fp32 sums then equal the bf16 image's times 2^k, bit for bit."""
    w = torch.cat([wh_f.detach().to(torch.float32).reshape(-1), wh_b.detach().to(torch.float32).reshape(-1)]).cpu()
    amax = float(w.abs().max())
    if amax == 0.0 or not math.isfinite(amax):
        return None
    k = int(math.floor(math.log2(32768.0 / amax)))
    ws = w * (2.0 ** k)
    if not bool((ws.to(torch.float16).to(torch.float32) == ws).all()) or float(ws.abs().max()) > 65504.0:
        return None
    nz = ws[ws != 0].abs()
    if nz.numel() and float(nz.min()) < 2.0 ** -14:      # a subnormal half would still be exact here, but fma_mix may flush it: stay in the normal range
        return None
    img = pack_lstm_wh(wh_f.detach().to(torch.float32) * (2.0 ** k), wh_b.detach().to(torch.float32) * (2.0 ** k), device, f16=True)
    return img, 2.0 ** -k


def pack_lstm_seq_wh(wh: torch.Tensor, device, f16: bool = False) -> RowMajor16:
    """Recurrent weights ``Wh`` [4H, H] (gate blocks i | f | g | o, the reference's layout) for ``lstm_seq``.  For H % 64 == 0 the rows are re-ordered by
    hidden unit (row 4 j + g): a 16-row tile of the step's GEMM then holds the four gates of four units and the whole step is ONE launch (gates in the
    GEMM's epilogue); other sizes keep the block order and the two-launch step.  ``MI355_LSTM_SEQ_FUSED=0`` keeps the block order everywhere (A/B)."""
    wh = wh.detach().to(torch.float32).cpu()
    h4, h = wh.shape
    assert h4 == 4 * h, wh.shape
    fused = h % 64 == 0 and os.environ.get("MI355_LSTM_SEQ_FUSED", "1") != "0"
    if fused:
        wh = wh.view(4, h, h).permute(1, 0, 2).reshape(4 * h, h).contiguous()
    rm = pack_rowmajor16(wh, None, device, f16=f16)
    rm.interleaved = fused
    return rm


def lstm_seq(xproj: torch.Tensor, wh: RowMajor16, out: torch.Tensor, h0: Optional[torch.Tensor] = None, c0: Optional[torch.Tensor] = None):
    """Unidirectional LSTM recurrence of any hidden size (``mi355_lstm_seq``): ``xproj`` [B, T, 4H] = x @ Wx^T + b (gate order i | f | g | o), ``wh`` the
    row-major 16-bit image of Wh [4H, H] (``pack_lstm_seq_wh``), ``out`` [B, T, H].  Returns (h_T, c_T)."""
    B, T, H4 = xproj.shape
    H = H4 // 4
    assert wh.n == H4 and wh.k == H and wh.scale is None and out.shape == (B, T, H) and xproj.stride(2) == 1 and out.stride(2) == 1
    h = torch.zeros((B, H), dtype=torch.float32, device=xproj.device) if h0 is None else h0.to(torch.float32).contiguous().clone()
    c = torch.zeros((B, H), dtype=torch.float32, device=xproj.device) if c0 is None else c0.to(torch.float32).contiguous().clone()
    pre = torch.empty((B, H4), dtype=torch.float32, device=xproj.device)
    h2 = torch.empty_like(h) if wh.interleaved else None
    _lib.call_struct("mi355_lstm_seq", "mi355_lstm_seq_args", _stream(), xproj=_ptr(xproj), xproj_bstride=xproj.stride(0), ld_xproj=xproj.stride(1),
                     wh=_ptr(wh.w), wdtype=wh.wdtype, h=_ptr(h), c=_ptr(c), pre=_ptr(pre), out=_ptr(out), out_bstride=out.stride(0), ld_out=out.stride(1),
                     B=B, T=T, H=H, gate_interleaved=int(wh.interleaved), h2=_ptr(h2))
    return h, c


# --------------------------------------------------------------------------------------- conv / linear
def conv_gemm(x: torch.Tensor, pc: PackedConv, y: torch.Tensor, *, dil: int = 1, pad: int = 0,
              lens_in: Optional[torch.Tensor] = None, lens_out: Optional[torch.Tensor] = None,
              lout: Optional[int] = None, pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
              pre_act: int = ACT_NONE, pre_slope: float = 0.0, pre_alpha: Optional[torch.Tensor] = None,
              post_act: int = ACT_NONE, post_slope: float = 0.0, res: Optional[torch.Tensor] = None,
              res_shift: int = 0, out_scale: float = 1.0, accumulate: bool = False,
              up: Optional[dict] = None, precision: int = 2, tile: int = 0,
              flat: Optional[dict] = None, use_bias: bool = True, stats: Optional[torch.Tensor] = None,
              pre_inv_beta: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None, x_off: int = 0,
              flatten: bool = False, pre_fq: Optional[torch.Tensor] = None, ext: Optional[torch.Tensor] = None,
              x_split: bool = False, y_split: bool = False):
    """y = epilogue(conv1d(prologue(x)))  -- see mi355_conv_gemm_args in the header.

    ``x_split`` / ``y_split``: the float32 tensors ``x`` / ``y`` hold SPLIT words (16-bit hi | lo << 16 of each value in the type of the launch's
    precision, 2 or 4: ``split16``, ``layernorm(split=...)`` or another launch's ``y_split``) -- the conversion is then done once by the producer
    instead of once per column tile of this launch.

    ``ext`` ([B, ceil(Lout / 64), Cout, 2] from ``new_ext``): the launch also leaves the per-block, per-channel (min, max) of what it stores, for
    ``fake_quant_extrema_from_partials`` -- only launches ``conv_ext_supported`` accepts (a quantising prologue on the wave-specialised kernel).

    ``flatten``: the caller states that this is a per-row linear layer (K == 1) whose padding rows (rows >= lens[b]) may be
    computed and written like any other row (nothing downstream reads them as valid): ``[B, L, C]`` operands whose items are
    row-contiguous are then handed over as ONE item of B*L rows, so the 128-row tiles are full instead of one partly
    filled tile per utterance (PL-BERT at T = 80: 62 % -> 100 % useful rows)."""
    if flatten and pc.k == 1 and up is None and flat is None and res_shift == 0 and stats is None and pre is None and pre_fq is None and x.shape[0] > 1:
        ts = [x, y] + ([res] if res is not None else [])
        if all(t.dim() == 3 and t.shape[:2] == x.shape[:2] and t.stride(0) == t.shape[1] * t.stride(1) for t in ts):
            fl = lambda t: t.as_strided((1, t.shape[0] * t.shape[1], t.shape[2]), (t.shape[0] * t.shape[1] * t.stride(1), t.stride(1), 1))
            return conv_gemm(fl(x), pc, fl(y), pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha, post_act=post_act,
                             post_slope=post_slope, res=None if res is None else fl(res), out_scale=out_scale, accumulate=accumulate,
                             precision=precision, tile=tile, use_bias=use_bias, pre_inv_beta=pre_inv_beta, colscale=colscale, x_off=x_off,
                             x_split=x_split, y_split=y_split)
    B, Lin, Cx, xbs, ldx = _nlc(x)
    By, Ly, Cy, ybs, ldy = _nlc(y)
    assert B == By
    if pc.mx:
        precision = 6 if pc.mx == 2 else 5  # the weight image decides
    elif pc.f16:
        if precision not in (3, 4):
            precision = 4 if precision in (5, 6) else 3  # fp16-packed weights only fit the fp16 MFMA paths (3 single, 4 hi+lo; mode 5 = hi+lo where no MX image exists)
    elif precision in (3, 4, 5, 6):
        raise _lib.Mi355Error("conv_gemm: precision 3 / 4 need weights packed with f16=True, precision 5 / 6 with mx=True / mx=2")
    kw = dict(x=_ptr(x), x_bstride=xbs, ldx=ldx, x_off=x_off, Cin=pc.cin, Lin=Lin, lens_in=_ptr(lens_in), flat_valid=0,
              w=_ptr(pc.w), Cout=pc.cout, K=pc.k, dil=dil, pad=pad, pre_act=pre_act, pre_slope=pre_slope,
              pre_alpha=_ptr(pre_alpha), bias=_ptr(pc.bias) if use_bias else None, post_act=post_act,
              post_slope=post_slope, out_scale=out_scale, accumulate=int(accumulate), y=_ptr(y), y_bstride=ybs, ldy=ldy,
              Lout=lout if lout is not None else Ly, lens_out=_ptr(lens_out), B=B, precision=precision, tile=tile,
              pre_inv_beta=_ptr(pre_inv_beta), post_colscale=_ptr(colscale))
    if x_split or y_split:
        if precision not in (2, 4):
            raise _lib.Mi355Error("conv_gemm: split activations exist for the hi + lo precisions 2 and 4 only")
        kw.update(x_split=precision if x_split else 0, y_split=precision if y_split else 0)
    if pre_fq is not None:  # [B, 2] from fake_quant_extrema (same prologue arguments): the prologue ends with the dynamic uint8 fake quantisation
        assert pre_fq.dtype == torch.float32 and pre_fq.shape == (B, 2) and pre_fq.is_contiguous()
        kw.update(pre_fq=_ptr(pre_fq))
    if flat is not None:  # flattened strided conv: taps are contiguous in memory (C_in small)
        kw.update(ldx=flat["ldx"], x_off=flat["x_off"], flat_valid=flat["channels"])
    else:
        assert Cx >= pc.cin or ldx >= pc.cin, (Cx, pc.cin)
    if pre is not None:
        sc, sh = pre
        assert sc.dim() == 2 and sc.stride(1) == 1 and sc.stride(0) == sh.stride(0)
        kw.update(pre_scale=_ptr(sc), pre_shift=_ptr(sh), pre_ld=sc.stride(0))
    if res is not None:
        _, _, _, rbs, ldr = _nlc(res)
        kw.update(res=_ptr(res), res_bstride=rbs, ldr=ldr, res_shift=res_shift)
    if up is not None:
        kw.update(up_s=up["s"], up_p=up["p"], up_cout=up["cout"], up_row_off=up.get("row_off", 0),
                  up_Lout=up["lout"], lens_up=_ptr(up.get("lens")))
    if ext is not None:
        assert ext.dtype == torch.float32 and ext.dim() == 4 and ext.is_contiguous() and ext.shape[0] == B and ext.shape[2] == pc.cout
        assert ext.shape[1] >= (kw["Lout"] + STATS_ROWS - 1) // STATS_ROWS
        kw.update(ext_partial=_ptr(ext), ext_bstride=ext.stride(0))
    if stats is not None:  # [B, ceil(Lout / 64), Cout, 2] float32: per-row-block (sum, M2) of the stored output
        assert stats.dtype == torch.float32 and stats.dim() == 4 and stats.is_contiguous()
        assert stats.shape[0] == B and stats.shape[1] >= (kw["Lout"] + STATS_ROWS - 1) // STATS_ROWS and stats.shape[2] == pc.cout
        kw.update(stats_partial=_ptr(stats), stats_bstride=stats.stride(0))
    if SPLIT_WS_BYTES and B * kw["Lout"] <= 65536:   # only small launches can split: do not even create the scratch for a large batch
        st = _stream()
        kw.update(split_ws=_conv_split_ws(x.device, st).data_ptr(), split_ws_bytes=SPLIT_WS_BYTES)
    if PROFILE is not None:
        # no host synchronisation here (a .item() on the ragged lengths would drain the queue before every launch: the start event would then
        # be stamped on an idle GPU and the interval would include the host's submission latency); the row count stays a device scalar and
        # profile_finalize() resolves it after the step
        rows = (kw["Lout"] * B) if lens_out is None else lens_out.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call_struct("mi355_conv_gemm", "mi355_conv_gemm_args", _stream(), **kw)
        e1.record()
        PROFILE.append((rows, pc.cin, pc.cout, pc.k, dil, int(res is not None) + int(bool(accumulate)), e0, e1))
        return y
    _lib.call_struct("mi355_conv_gemm", "mi355_conv_gemm_args", _stream(), **kw)
    return y


def adain_coef(x: torch.Tensor, gb: Optional[torch.Tensor], lens: Optional[torch.Tensor] = None, eps: float = 1e-5, sums: Optional[torch.Tensor] = None,
               reuse: bool = False):
    """Instance-norm stats of ``x`` [B, L, C] + AdaIN coefficients -> (scale, shift) each [B, C padded to 32].
    ``sums`` (float64 [B * C * 2]) + ``reuse=True``: the statistics of this same ``x`` are already in ``sums`` (an earlier call): only the coefficients
    for another ``gb`` are computed -- one statistics pass for all blocks that normalise the same input."""
    B, L, C, xbs, ldx = _nlc(x)
    cp = round_up(C, 32)
    if sums is None:
        assert not reuse
        sums = torch.empty(B * C * 2, dtype=torch.float64, device=x.device)
    scale = torch.empty((B, cp), dtype=torch.float32, device=x.device)
    shift = torch.empty((B, cp), dtype=torch.float32, device=x.device)
    _lib.call_struct("mi355_adain_coef", "mi355_adain_coef_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L,
                     lens=_ptr(lens), B=B, sums=_ptr(sums), gb=_ptr(gb), gb_ld=0 if gb is None else gb.stride(0), eps=eps,
                     scale=_ptr(scale), shift=_ptr(shift), out_ld=cp, reuse_sums=int(reuse))
    return scale, shift


STATS_ROWS = 64  # MI355_STATS_ROWS

# scratch of mi355_conv_gemm's split-K path (launches of few output tiles: one utterance per call), one per (device, stream): launches on a stream
# are ordered, so consecutive convs can share it.  A fixed 96 MB (MI355_CONV_SPLIT_WS_MB): the dispatcher caps the group count by what fits, so a
# smaller workspace only means fewer groups.  NOTE (round-3 advisor): whether a launch splits depends on its tile count B * L_out, so one utterance
# takes a different fp32 summation order alone than inside a batch -- deterministic per shape, equal across shapes only to rounding; code that
# asserts integer results across batch sizes (durations, token ids) must do so under a margin (tests/test_kokoro_gpu.py, tests/_margin.py)
SPLIT_WS_BYTES = int(os.environ.get("MI355_CONV_SPLIT_WS_MB", "96")) << 20
_CONV_SPLIT_WS = {}


def _conv_split_ws(device: torch.device, stream: int) -> torch.Tensor:
    key = (device.index, stream)
    ws = _CONV_SPLIT_WS.get(key)
    if ws is None:
        ws = _CONV_SPLIT_WS[key] = torch.empty(SPLIT_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


def new_stats(B: int, L: int, C: int, device) -> torch.Tensor:
    """Buffer for the fused instance-norm statistics of a conv output [B, L, C] (see conv_gemm(stats=...))."""
    return torch.empty((B, (L + STATS_ROWS - 1) // STATS_ROWS, C, 2), dtype=torch.float32, device=device)


def new_ext(B: int, L: int, C: int, device) -> torch.Tensor:
    """Buffer for the per-block extrema a conv launch leaves (``conv_gemm(ext=...)``): [B, ceil(L / 64), C, 2] = (min, max)."""
    return torch.empty((B, (L + STATS_ROWS - 1) // STATS_ROWS, C, 2), dtype=torch.float32, device=device)


def conv_ext_supported(x: torch.Tensor, pc: PackedConv, *, B: int, lout: int, dil: int = 1, pre_act: int = ACT_NONE, post_act: int = ACT_NONE,
                       up: Optional[dict] = None, flat: Optional[dict] = None, precision: int = 2, min_tiles: int = 128) -> bool:
    """Will ``conv_gemm(..., pre_fq=..., ext=...)`` of this shape be taken by the kernel that writes extrema partials -- AND would the dispatcher have
    chosen that kernel anyway (at least ``min_tiles`` 128-row tiles: the rule of ``mi355_conv_gemm``'s automatic choice), so that asking for the
    partials does not change which kernel computes the conv?"""
    if up is not None or flat is not None or pc.f16 or pc.mx or precision != 2 or pc.k == 1:
        return False
    _, _, _, xbs, ldx = _nlc(x)
    bn = 64 if pc.cout <= 64 else 128
    if B * ((lout + 127) // 128) * ((pc.cout + bn - 1) // bn) < min_tiles:
        return False
    return bool(_lib.call_struct_ret("mi355_conv_gemm_ext_supported", "mi355_conv_gemm_args", x=_ptr(x), x_bstride=xbs, ldx=ldx, Cin=pc.cin, Lin=x.shape[1],
                                     w=_ptr(pc.w), Cout=pc.cout, K=pc.k, dil=dil, pre_act=pre_act, post_act=post_act, Lout=lout, B=B, precision=2,
                                     pre_alpha=_ptr(pc.w), pre_fq=_ptr(pc.w), y=_ptr(pc.w)))   # the probe reads shapes, flags and x's alignment; the other pointers only have to be set


def fake_quant_extrema_from_partials(ext: torch.Tensor, L: int, *, lens=None, pre=None, pre_act: int = ACT_NONE, pre_slope: float = 0.0,
                                     pre_alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[B, 2]`` = {-min, max} of ``act(scale * x + shift)`` per utterance from the per-block, per-channel (min, max) of x that the conv producing x
    left (``conv_gemm(ext=...)``): what ``fake_quant_extrema`` computes by reading x again."""
    B, nblk, C, two = ext.shape
    assert two == 2 and ext.is_contiguous() and nblk >= (L + STATS_ROWS - 1) // STATS_ROWS
    mm = torch.empty((B, 2), dtype=torch.float32, device=ext.device)
    sc, sh = pre if pre is not None else (None, None)
    _lib.call_struct("mi355_fake_quant_extrema_from_partials", "mi355_fake_quant_args", _stream(), x=_ptr(ext), x_bstride=ext.stride(0), ldx=2 * C, C=C, L=L,
                     lens=_ptr(lens), B=B, pre_scale=_ptr(sc), pre_shift=_ptr(sh), pre_ld=sc.stride(0) if sc is not None else 0, pre_act=pre_act,
                     pre_slope=pre_slope, pre_alpha=_ptr(pre_alpha), minmax=_ptr(mm))
    return mm


def adain_from_partials(stats: torch.Tensor, L: int, gb: Optional[torch.Tensor], lens: Optional[torch.Tensor] = None, eps: float = 1e-5):
    """AdaIN (scale, shift) from the per-row-block partial statistics a conv_gemm epilogue wrote."""
    B, _, C, _ = stats.shape
    cp = round_up(C, 32)
    scale = torch.empty((B, cp), dtype=torch.float32, device=stats.device)
    shift = torch.empty((B, cp), dtype=torch.float32, device=stats.device)
    _lib.call_struct("mi355_adain_from_partials", "mi355_adain_partials_args", _stream(), partials=_ptr(stats), bstride=stats.stride(0),
                     C=C, L=L, lens=_ptr(lens), B=B, gb=_ptr(gb), gb_ld=0 if gb is None else gb.stride(0), eps=eps,
                     scale=_ptr(scale), shift=_ptr(shift), out_ld=cp)
    return scale, shift


def layernorm(x: torch.Tensor, y: torch.Tensor, *, weight=None, bias=None, ada_gb=None, res=None, eps=1e-5,
              lens=None, post_act=ACT_NONE, post_slope=0.0, split: int = 0):
    """``split`` = 2 / 4: ``y`` receives SPLIT words (``conv_gemm(..., x_split=True)`` at that precision reads them) instead of floats."""
    B, L, C, xbs, ldx = _nlc(x)
    _, _, _, ybs, ldy = _nlc(y)
    kw = dict(x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L, lens=_ptr(lens), B=B, weight=_ptr(weight), bias=_ptr(bias),
              ada_gb=_ptr(ada_gb), ada_ld=0 if ada_gb is None else ada_gb.stride(0), eps=eps, post_act=post_act,
              post_slope=post_slope, y=_ptr(y), y_bstride=ybs, ldy=ldy, y_split=int(split))
    if res is not None:
        _, _, _, rbs, ldr = _nlc(res)
        kw.update(res=_ptr(res), res_bstride=rbs, ldr=ldr)
    _lib.call_struct("mi355_layernorm", "mi355_layernorm_args", _stream(), **kw)
    return y


def split16(x: torch.Tensor, fmt: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> SPLIT words (16-bit hi | lo << 16; ``fmt`` 2 = bfloat16, 4 = IEEE half) of a contiguous float32 tensor, for ``conv_gemm(..., x_split=True)`` at
    precision ``fmt``; ``y`` defaults to a fresh tensor (``y is x``: in place).  The words travel in a float32 tensor: read them with ``.view(torch.int32)``."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0
    if y is None:
        y = torch.empty_like(x)
    assert y.dtype == torch.float32 and y.is_contiguous() and y.numel() == x.numel()
    lib = _lib.load()
    _lib.check(lib.mi355_split16(_ptr(x), _ptr(y), x.numel(), int(fmt), _stream()), "mi355_split16")
    return y


def lstm_bidir(xp: torch.Tensor, wh: torch.Tensor, H: int, out: torch.Tensor, lens=None, quant_h: bool = False, wh_f16: bool = False,
               wh_scale: float = 0.0):
    B, L, _, xbs, ldxp = _nlc(xp)
    _, _, _, obs, ldo = _nlc(out)
    _lib.call_struct("mi355_lstm_bidir", "mi355_lstm_args", _stream(), xp=_ptr(xp), xp_bstride=xbs, ldxp=ldxp, wh=_ptr(wh),
                     H=H, L=L, lens=_ptr(lens), B=B, out=_ptr(out), out_bstride=obs, ldo=ldo, quant_h=int(bool(quant_h)), wh_f16=int(bool(wh_f16)),
                     wh_scale=float(wh_scale))
    return out


def fake_quant_u8(x: torch.Tensor, y: Optional[torch.Tensor] = None, *, lens=None, pre=None, pre_act: int = ACT_NONE, pre_slope: float = 0.0,
                  pre_alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``fake_quant_dynamic_u8`` (tts/models/kitten_tts/quant.py) of ``act(scale * x + shift)`` per utterance: x [B, L, C] -> y (a fresh
    tensor unless given; ``y is x`` quantises in place)."""
    B, L, C, xbs, ldx = _nlc(x)
    if y is None:  # channel-padded like every conv input of the engines (rows past lens[b] and the pad columns stay zero)
        y = torch.zeros((B, L, round_up(C, 32)), dtype=torch.float32, device=x.device)[:, :, :C]
    _, _, _, ybs, ldy = _nlc(y)
    mm = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    sc, sh = pre if pre is not None else (None, None)
    _lib.call_struct("mi355_fake_quant_u8", "mi355_fake_quant_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L, lens=_ptr(lens), B=B,
                     pre_scale=_ptr(sc), pre_shift=_ptr(sh), pre_ld=sc.stride(0) if sc is not None else 0, pre_act=pre_act, pre_slope=pre_slope,
                     pre_alpha=_ptr(pre_alpha), y=_ptr(y), y_bstride=ybs, ldy=ldy, minmax=_ptr(mm))
    return y


def rvq_encode(x: torch.Tensor, tables: torch.Tensor, tables_t: torch.Tensor, c2: torch.Tensor, margins: bool = False):
    """Residual VQ encode of the projected frames ``x`` [rows, D] over ``tables`` [n, bins, D] (+ transposed copy [n, D, bins] and |e|^2 / 2 [n, bins]):
    codes int32 [rows, n] (and the best-vs-second score gaps when ``margins``)."""
    assert x.dim() == 2 and x.stride(1) == 1 and tables.is_contiguous() and tables_t.is_contiguous() and c2.is_contiguous()
    rows, D = x.shape
    n, bins, D2 = tables.shape
    assert D2 == D and tuple(tables_t.shape) == (n, D, bins) and tuple(c2.shape) == (n, bins)
    codes = torch.empty((rows, n), dtype=torch.int32, device=x.device)
    mg = torch.empty((rows, n), dtype=torch.float32, device=x.device) if margins else None
    _lib.call_struct("mi355_rvq_encode", "mi355_rvq_encode_args", _stream(), x=_ptr(x), rows=rows, ldx=x.stride(0), D=D, tables=_ptr(tables), tables_t=_ptr(tables_t),
                     c2=_ptr(c2), bins=bins, n_layers=n, codes=_ptr(codes), ld_codes=n, margins=_ptr(mg))
    return (codes, mg) if margins else codes


def ecapa_rows(x: torch.Tensor, y: torch.Tensor, *, pad: int = 0, gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
               pre_tanh: bool = False) -> torch.Tensor:
    """``y[b, r] = f(x[b, s]) * sigmoid(gate[b]) + res[b, s]``, ``s = reflect(r - pad)``: x / res [B, T, C] (channel slices of wider buffers are fine),
    gate [B, C] logits, y [B, T + 2 * pad, C] -- see mi355_ecapa_rows_args."""
    B, T, C, xbs, ldx = _nlc(x)
    By, Ty, Cy, ybs, ldy = _nlc(y)
    assert (By, Ty, Cy) == (B, T + 2 * pad, C), ((By, Ty, Cy), (B, T, C), pad)
    kw = dict(x=_ptr(x), x_bstride=xbs, ldx=ldx, pre_tanh=int(pre_tanh), y=_ptr(y), y_bstride=ybs, ldy=ldy, B=B, T=T, C=C, pad=pad)
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.shape == (B, C) and gate.stride(1) == 1
        kw.update(gate=_ptr(gate), gate_ld=gate.stride(0))
    if res is not None:
        Br, Tr, Cr, rbs, ldr = _nlc(res)
        assert (Br, Tr, Cr) == (B, T, C)
        kw.update(res=_ptr(res), res_bstride=rbs, ldr=ldr)
    _lib.call_struct("mi355_ecapa_rows", "mi355_ecapa_rows_args", _stream(), **kw)
    return y


def time_moments(x: torch.Tensor, eps: float = 0.0, want_std: bool = True):
    """(mean [B, C], sqrt(biased variance + eps) [B, C] or None) over the time axis of ``x`` [B, T, C]."""
    B, T, C, xbs, ldx = _nlc(x)
    mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
    std = torch.empty((B, C), dtype=torch.float32, device=x.device) if want_std else None
    _lib.call_struct("mi355_time_moments", "mi355_time_moments_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, B=B, T=T, C=C, eps=eps,
                     mean=_ptr(mean), std=_ptr(std), out_ld=C)
    return mean, std


def attentive_pool(x: torch.Tensor, logits: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """Softmax over time of ``logits`` per channel; [B, 2C] = (weighted mean | sqrt(max(weighted variance, eps))) of ``x`` -- mi355_attentive_pool_args."""
    B, T, C, xbs, ldx = _nlc(x)
    Bl, Tl, Cl, lbs, ldl = _nlc(logits)
    assert (Bl, Tl, Cl) == (B, T, C)
    out = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
    _lib.call_struct("mi355_attentive_pool", "mi355_attentive_pool_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, logits=_ptr(logits), l_bstride=lbs,
                     ldl=ldl, B=B, T=T, C=C, eps=eps, out=_ptr(out), out_ld=2 * C)
    return out


def fake_quant_extrema(x: torch.Tensor, *, lens=None, pre=None, pre_act: int = ACT_NONE, pre_slope: float = 0.0,
                       pre_alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Extrema pass alone: ``[B, 2]`` = {-min, max} of ``act(scale * x + shift)`` per utterance (joined with 0) for ``conv_gemm(pre_fq=...)``,
    which quantises inside its prologue -- the quantised tensor is never written."""
    B, L, C, xbs, ldx = _nlc(x)
    mm = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    sc, sh = pre if pre is not None else (None, None)
    _lib.call_struct("mi355_fake_quant_extrema", "mi355_fake_quant_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L, lens=_ptr(lens), B=B,
                     pre_scale=_ptr(sc), pre_shift=_ptr(sh), pre_ld=sc.stride(0) if sc is not None else 0, pre_act=pre_act, pre_slope=pre_slope,
                     pre_alpha=_ptr(pre_alpha), minmax=_ptr(mm))
    return mm


def attention(qkv: torch.Tensor, heads: int, dh: int, out: torch.Tensor, lens=None):
    B, T, _, bs, ld = _nlc(qkv)
    _, _, _, obs, ldo = _nlc(out)
    _lib.call_struct("mi355_attention", "mi355_attention_args", _stream(), qkv=_ptr(qkv), bstride=bs, ld=ld, heads=heads,
                     dh=dh, T=T, lens=_ptr(lens), B=B, out=_ptr(out), out_bstride=obs, ldo=ldo)
    return out


_SPLIT_WS = {}  # (device index, stream) -> (float workspace, zeroed int32 tickets) of the key-split decode attention


def attn_split_workspace(device, n_records: int, dh: int):
    """Persistent workspace for mi355_flash_attn_args.split_ws / split_cnt: ``n_records`` = B * heads * Tq.  The tickets must be zero before
    the first launch and every launch leaves them zero, so one allocation per (device, stream) serves all calls."""
    key = (torch.device(device).index or 0, _stream())
    need_ws, need_cnt = n_records * 8 * (dh + 2), n_records
    cur = _SPLIT_WS.get(key)
    if cur is None or cur[0].numel() < need_ws or cur[1].numel() < need_cnt:
        cur = (torch.empty(max(need_ws, 1 << 16), dtype=torch.float32, device=device),
               torch.zeros(max(need_cnt, 1 << 10), dtype=torch.int32, device=device))
        _SPLIT_WS[key] = cur
    return cur


_GEMV_SPLIT_WS = {}


def gemv_split_workspace(device, n: int, k: int):
    """(float32 scratch, zeroed int32 tickets) for mi355_gemv_args.split_ws / split_cnt, grown on demand and shared by every launch on ``device``
    (launches on one stream are ordered; the tickets are left zero by every call)."""
    tiles, nch = (n + 15) // 16, (k + 2047) // 2048
    need = tiles * nch * 256
    key = str(device)
    cur = _GEMV_SPLIT_WS.get(key)
    if cur is None:  # allocated ONCE per device and never re-allocated: C descriptors keep raw pointers to it
        cur = (torch.empty(1 << 22, dtype=torch.float32, device=device), torch.zeros(1 << 13, dtype=torch.int32, device=device))
        _GEMV_SPLIT_WS[key] = cur
    if need > cur[0].numel() or tiles > cur[1].numel():
        raise _lib.Mi355Error(f"gemv split workspace too small for N={n}, K={k}")
    return cur


KV_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}  # MI355_KV_F32 / MI355_KV_BF16 / MI355_KV_F16


def kv_head_major16(kv: torch.Tensor, groups: int, heads: int, dh: int, dtype: torch.dtype) -> torch.Tensor:
    """float32 ``kv[B, T, >= groups * heads * dh]`` (k | v projection rows) -> ``[groups, B, heads, T, dh]`` in ``dtype`` (float16 / bfloat16): the head-major 16-bit
    blocks the decode-step attention streams, in one pass (``mi355_kv_head_major16``)."""
    B, T, C, bs, ld = _nlc(kv)
    assert C >= groups * heads * dh and dtype in (torch.float16, torch.bfloat16)
    out = torch.empty((groups, B, heads, T, dh), dtype=dtype, device=kv.device)
    lib = _lib.load()
    _lib.check(lib.mi355_kv_head_major16(_ptr(kv), bs, ld, B, T, groups, heads, dh, _ptr(out), KV_DTYPES[dtype], _stream()), "mi355_kv_head_major16")
    return out


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, heads: int, kv_heads: Optional[int] = None,
                    dh: int, scale: Optional[float] = None, causal: bool = False, window: int = 0, lens_q=None, lens_k=None,
                    mode: int = 0, k_start=None, head_major: bool = False, nsplit: int = 0, fused: Optional[dict] = None):
    """softmax(scale * q k^T + visibility) v.  q/out [B, Tq, >= heads*dh], k/v [B, Tk, >= kv_heads*dh] channels-last views
    (a KV cache is just the buffer k / v point into); see mi355_flash_attn_args for the visibility rule.
    ``fused`` (single-query decode step): dict(new_k, new_v [B, kv_heads * dh] raw projections of the new position, q_norm_w, k_norm_w, eps,
    cos, sin [rows, dh / 2], rope_mode, pos) -- q / k norms, rotary embedding and the cache store of row Tk - 1 happen inside the attention kernel."""
    B, Tq, _, qbs, ldq = _nlc(q)
    _, To, _, obs, ldo = _nlc(out)
    khs = vhs = 0
    if head_major:  # k / v [B, kv_heads, Tk, dh]: a head's keys contiguous (the layout for long key ranges: one L2 channel sweep per head)
        assert k.dim() == 4 and v.dim() == 4 and k.stride(3) == 1 and v.stride(3) == 1 and k.shape == v.shape
        Bk, _, Tk, _ = k.shape
        Tv, kbs, khs, ldk, vbs, vhs, ldv = Tk, k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2)
    else:
        assert k.dim() == 3 and v.dim() == 3 and k.stride(2) == 1 and v.stride(2) == 1 and k.is_cuda and v.is_cuda
        Bk, Tk, kbs, ldk = k.shape[0], k.shape[1], k.stride(0), k.stride(1)
        Tv, vbs, ldv = v.shape[1], v.stride(0), v.stride(1)
    assert Bk == B and Tv == Tk and To == Tq
    assert k.dtype == v.dtype and k.dtype in KV_DTYPES, "k / v must both be float32, bfloat16 or float16"
    fkw = {}
    if fused is not None:
        nk, nv = fused["new_k"], fused["new_v"]
        assert nk.dim() == 2 and nv.dim() == 2 and nk.stride(1) == 1 and nv.stride(1) == 1 and nk.stride(0) == nv.stride(0) and nk.dtype == torch.float32
        cos, sin = fused.get("cos"), fused.get("sin")
        fkw = dict(new_k=_ptr(nk), new_v=_ptr(nv), new_bstride=nk.stride(0), q_norm_w=_ptr(fused.get("q_norm_w")), k_norm_w=_ptr(fused.get("k_norm_w")),
                   norm_eps=fused.get("eps", 1e-6), rope_cos=_ptr(cos), rope_sin=_ptr(sin), rope_rows=0 if cos is None else cos.shape[0],
                   rope_mode=fused.get("rope_mode", 0), rope_pos=fused.get("pos", Tk - 1))
    sws = scnt = None
    if nsplit > 1 and Tq <= 8 and mode != 1:
        sws, scnt = attn_split_workspace(q.device, B * heads * Tq, dh)  # key-split decode (flash-decoding), opt-in: measured slower than the unsplit kernel
    _lib.call_struct("mi355_flash_attention", "mi355_flash_attn_args", _stream(), q=_ptr(q), q_bstride=qbs, ldq=ldq, k=_ptr(k),
                     k_bstride=kbs, ldk=ldk, v=_ptr(v), v_bstride=vbs, ldv=ldv, heads=heads, kv_heads=kv_heads or heads, dh=dh,
                     Tq=Tq, Tk=Tk, lens_q=_ptr(lens_q), lens_k=_ptr(lens_k), causal=int(causal), window=window,
                     scale=(1.0 / math.sqrt(dh)) if scale is None else scale, B=B, mode=mode, out=_ptr(out), out_bstride=obs, ldo=ldo,
                     k_start=_ptr(k_start), k_hstride=khs, v_hstride=vhs, split_ws=_ptr(sws), split_cnt=_ptr(scnt), nsplit=nsplit,
                     kv_dtype=KV_DTYPES[k.dtype], **fkw)
    return out


def whisper_greedy_step(logits: torch.Tensor, tokens: torch.Tensor, n: int, sample_begin: int, sum_logprobs: torch.Tensor, *,
                        V: Optional[int] = None, suppress_mask=None, blank_ids=None, timestamp_rules: bool = False,
                        timestamp_begin: int = 0, eot: int = 0, no_timestamps: int = -1, max_initial_timestamp_index: int = -1,
                        gumbel=None, temperature: float = 0.0, filtered=None, forced_next=None, split_ws=None):
    """One decode step of decoding.py's filter chain + GreedyDecoder.update, in place on ``tokens[:, n]`` / ``sum_logprobs``."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype == torch.float32
    assert tokens.dtype == torch.int32 and tokens.dim() == 2 and tokens.stride(1) == 1
    B = logits.shape[0]
    _lib.call_struct("mi355_whisper_greedy_step", "mi355_whisper_step_args", _stream(), logits=_ptr(logits), ld=logits.stride(0),
                     V=V or logits.shape[1], B=B, tokens=_ptr(tokens), tokens_ld=tokens.stride(0), n=n, sample_begin=sample_begin,
                     suppress_mask=_ptr(suppress_mask), blank_ids=_ptr(blank_ids), n_blank=0 if blank_ids is None else blank_ids.numel(),
                     timestamp_rules=int(timestamp_rules), timestamp_begin=timestamp_begin, eot=eot, no_timestamps=no_timestamps,
                     max_initial_timestamp_index=max_initial_timestamp_index, gumbel=_ptr(gumbel), temperature=temperature,
                     sum_logprobs=_ptr(sum_logprobs), filtered=_ptr(filtered), forced_next=_ptr(forced_next),
                     split_ws=None if split_ws is None else _ptr(split_ws[0]), split_cnt=None if split_ws is None else _ptr(split_ws[1]))


def whisper_step_workspace(device, B: int):
    """(float32 [B * 16 * 12], int32 [B] zeroed): the scratch of the multi-workgroup decode-rules step (mi355_whisper_step_args.split_ws / split_cnt)."""
    return (torch.empty(B * 16 * 12, dtype=torch.float32, device=device), torch.zeros(B, dtype=torch.int32, device=device))


def softmax_prob_at(logits: torch.Tensor, token: int, V: Optional[int] = None) -> torch.Tensor:
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype == torch.float32
    out = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    lib = _lib.load()
    rc = lib.mi355_softmax_prob_at(_ptr(logits), logits.stride(0), V or logits.shape[1], logits.shape[0], token, _ptr(out), _stream())
    _lib.check(rc, "mi355_softmax_prob_at")
    return out


def rmsnorm(x: torch.Tensor, y: torch.Tensor, weight: Optional[torch.Tensor], eps: float = 1e-6, lens=None):
    B, L, C, xbs, ldx = _nlc(x)
    _, _, _, ybs, ldy = _nlc(y)
    _lib.call_struct("mi355_rmsnorm", "mi355_rmsnorm_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L, lens=_ptr(lens), B=B,
                     weight=_ptr(weight), eps=eps, y=_ptr(y), y_bstride=ybs, ldy=ldy)
    return y


def head_norm_rope(x: torch.Tensor, y: torch.Tensor, *, heads: int, dh: int, norm_weight: Optional[torch.Tensor] = None, eps: float = 1e-6,
                   cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None, pos0: int = 0,
                   interleaved: bool = False, lens=None, second: Optional[tuple] = None, pos_sub: Optional[torch.Tensor] = None):
    """Per-head RMSNorm (optional) + RoPE (optional) of ``heads`` heads starting at column 0 of ``x`` into ``y`` (may be a cache slot).
    ``pos_sub`` int32 [B]: left padding of each row, subtracted from its positions (its first real token is position 0)."""
    B, L, _, xbs, ldx = _nlc(x)
    _, _, _, ybs, ldy = _nlc(y)
    if cos is not None:
        assert cos.dim() == 2 and cos.shape[1] == dh // 2 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == sin.shape
    if pos is not None:
        assert pos.dtype == torch.int32 and pos.dim() == 2 and pos.stride(1) == 1
    elif cos is not None and pos0 + L > cos.shape[0]:
        raise ValueError(f"head_norm_rope: positions {pos0}..{pos0 + L - 1} run past the {cos.shape[0]}-row rotary tables")
    kw = {}
    if second is not None:  # (x2, y2, heads2, norm_weight2): e.g. the k heads, rotated straight into their KV-cache slot
        x2, y2, heads2, nw2 = second
        B2, L2, _, x2bs, ldx2 = _nlc(x2)
        _, _, _, y2bs, ldy2 = _nlc(y2)
        assert B2 == B and L2 == L
        kw = dict(x2=_ptr(x2), x2_bstride=x2bs, ldx2=ldx2, heads2=heads2, norm_weight2=_ptr(nw2), y2=_ptr(y2), y2_bstride=y2bs, ldy2=ldy2)
    _lib.call_struct("mi355_head_norm_rope", "mi355_head_rope_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, heads=heads, dh=dh, L=L,
                     lens=_ptr(lens), B=B, norm_weight=_ptr(norm_weight), eps=eps, cos_table=_ptr(cos), sin_table=_ptr(sin), pos=_ptr(pos),
                     pos_ld=0 if pos is None else pos.stride(0), pos0=pos0, pos_sub=_ptr(pos_sub), rope_rows=0 if cos is None else cos.shape[0],
                     rope_mode=int(interleaved), y=_ptr(y), y_bstride=ybs, ldy=ldy, **kw)
    return y


def swiglu(x: torch.Tensor, y: torch.Tensor):
    """y[..., i] = silu(x[..., 2i]) * x[..., 2i+1] on contiguous-row 3-D views [B, L, 2I] -> [B, L, I] (both dense over B, L)."""
    B, L, C2, xbs, ldx = _nlc(x)
    _, _, I, ybs, ldy = _nlc(y)
    assert C2 == 2 * I and xbs == L * ldx and ybs == L * ldy
    lib = _lib.load()
    rc = lib.mi355_swiglu(_ptr(x), ldx, _ptr(y), ldy, B * L, I, _stream())
    _lib.check(rc, "mi355_swiglu")
    return y


def embed_sum(table: torch.Tensor, ids: torch.Tensor, y: torch.Tensor, *, slot_offset: Optional[torch.Tensor] = None,
              add: Optional[torch.Tensor] = None, scale: float = 1.0, lens=None):
    """y[b, l] = scale * (add[b, l] + sum_q table[slot_offset[q] + ids[b, l, q]]); ``ids`` int32 [B, L, Q] (any strides)."""
    B, L, C, ybs, ldy = _nlc(y)
    assert ids.dtype == torch.int32 and ids.dim() == 3 and ids.shape[0] == B and ids.shape[1] == L
    assert table.dim() == 2 and table.stride(1) == 1 and table.dtype == torch.float32
    kw = dict(table=_ptr(table), ld_table=table.stride(0), slot_offset=_ptr(slot_offset), ids=_ptr(ids), ids_bstride=ids.stride(0),
              ids_ld=ids.stride(1), ids_qstride=ids.stride(2), Q=ids.shape[2], scale=scale, C=C, L=L, lens=_ptr(lens), B=B, y=_ptr(y),
              y_bstride=ybs, ldy=ldy)
    if add is not None:
        _, _, _, abs_, ald = _nlc(add)
        kw.update(add=_ptr(add), add_bstride=abs_, add_ld=ald)
    _lib.call_struct("mi355_embed_sum", "mi355_embed_sum_args", _stream(), **kw)
    return y


def dwconv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor, *, pad: int = 0, stride: int = 1,
           transpose: bool = False, lens_in=None, dil: int = 1, pre_alpha: Optional[torch.Tensor] = None, pre_inv: Optional[torch.Tensor] = None):
    """Depthwise conv / conv_transpose, channels-last; ``w`` [C, K] float32.  ``dil`` and the per-channel Snake prologue
    (``pre_alpha``, ``pre_inv`` = 1 / (alpha + 1e-9), each [>= C]) apply to the plain conv."""
    B, Lin, C, xbs, ldx = _nlc(x)
    _, Lout, _, ybs, ldy = _nlc(y)
    assert w.dim() == 2 and w.shape[0] == C and w.is_contiguous() and w.dtype == torch.float32
    _lib.call_struct("mi355_dwconv", "mi355_dwconv_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, Lin=Lin, lens_in=_ptr(lens_in),
                     w=_ptr(w), bias=_ptr(bias), C=C, K=w.shape[1], pad=pad, stride=stride, transpose=int(transpose), B=B, y=_ptr(y),
                     y_bstride=ybs, ldy=ldy, Lout=Lout, dil=dil, pre_alpha=_ptr(pre_alpha), pre_inv=_ptr(pre_inv))
    return y


def sample(logits: torch.Tensor, out: torch.Tensor, *, V: Optional[int] = None, suppress_mask=None, history=None, n_hist: int = 0,
           hist_len=None, repetition_penalty: float = 1.0, temperature: float = 0.0, top_k: int = 0, top_p: float = 1.0, min_p: float = 0.0,
           gumbel=None, done=None, done_token: int = 0, filtered=None):
    """Samples one token per row of ``logits`` [B, ld] into ``out`` (int32 view with B elements, any stride)."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype == torch.float32 and out.dtype == torch.int32
    B = logits.shape[0]
    out_ld = out.stride(0) if out.dim() >= 1 and out.shape[0] == B and B > 1 else 1
    kw = dict(logits=_ptr(logits), ld=logits.stride(0), V=V or logits.shape[1], B=B, suppress_mask=_ptr(suppress_mask),
              repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p, min_p=min_p, gumbel=_ptr(gumbel),
              done=_ptr(done), done_token=done_token, out=_ptr(out), out_ld=out_ld, filtered=_ptr(filtered))
    if history is not None:
        assert history.dtype == torch.int32 and history.dim() == 2 and history.stride(1) == 1
        kw.update(history=_ptr(history), hist_ld=history.stride(0), n_hist=n_hist, hist_len=_ptr(hist_len))
    if gumbel is not None:
        assert gumbel.stride(0) == logits.stride(0)
    if filtered is not None:
        assert filtered.stride(0) == logits.stride(0)
    _lib.call_struct("mi355_sample", "mi355_sample_args", _stream(), **kw)
    return out


def gather_rows(table: torch.Tensor, idx: torch.Tensor, y: torch.Tensor, *, per_batch: bool = False, pos_table=None,
                add_row=None, lens=None):
    B, L, C, ybs, ldy = _nlc(y)
    assert idx.dtype == torch.int32 and idx.dim() == 2 and idx.stride(1) == 1
    if per_batch:
        assert table.dim() == 3 and table.stride(2) == 1
        tbs, ldt = table.stride(0), table.stride(1)
    else:
        assert table.dim() == 2 and table.stride(1) == 1
        tbs, ldt = 0, table.stride(0)
    _lib.call_struct("mi355_gather_rows", "mi355_gather_rows_args", _stream(), table=_ptr(table), ld_table=ldt,
                     table_bstride=tbs, pos_table=_ptr(pos_table), ld_pos=0 if pos_table is None else pos_table.stride(0),
                     add_row=_ptr(add_row), idx=_ptr(idx), idx_ld=idx.stride(0), C=C, L=L, lens=_ptr(lens), B=B, y=_ptr(y),
                     y_bstride=ybs, ldy=ldy)
    return y


def broadcast_rows(v: torch.Tensor, y: torch.Tensor, lens=None):
    B, L, C, ybs, ldy = _nlc(y)
    assert v.dim() == 2 and v.stride(1) == 1
    lib = _lib.load()
    rc = lib.mi355_broadcast_rows(_ptr(v), v.stride(0), C, _ptr(y), ybs, ldy, L, _ptr(lens), B, _stream())
    _lib.check(rc, "mi355_broadcast_rows")
    return y


def duration_align(logits: Optional[torch.Tensor], T: int, B: int, speed: float, idx_cap: int, device, lens=None,
                   forced: Optional[torch.Tensor] = None, bins: int = 50, max_frames: int = 0):
    dur = torch.empty((B, T), dtype=torch.int32, device=device)
    raw = torch.zeros((B, T), dtype=torch.float32, device=device)
    frames = torch.empty((B,), dtype=torch.int32, device=device)
    idx = torch.zeros((B, idx_cap), dtype=torch.int32, device=device)
    kw = dict(T=T, lens=_ptr(lens), B=B, speed=speed, forced_dur=_ptr(forced), dur=_ptr(dur), dur_raw=_ptr(raw),
              frames=_ptr(frames), idx=_ptr(idx), idx_ld=idx_cap, bins=bins, max_frames=max_frames)
    if logits is not None:
        _, _, _, bs, ld = _nlc(logits)
        kw.update(logits=_ptr(logits), bstride=bs, ld=ld)
    _lib.call_struct("mi355_duration_align", "mi355_duration_args", _stream(), **kw)
    return dur, raw, frames, idx


def adain_pool_up2(x: torch.Tensor, scale, shift, slope: float, w: torch.Tensor, bias: torch.Tensor, y: torch.Tensor, lens=None):
    B, L, C, xbs, ldx = _nlc(x)
    _, _, _, ybs, ldy = _nlc(y)
    _lib.call_struct("mi355_adain_pool_up2", "mi355_pool_up2_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L,
                     lens=_ptr(lens), B=B, scale=_ptr(scale), shift=_ptr(shift), pre_ld=scale.stride(0), slope=slope,
                     w=_ptr(w), bias=_ptr(bias), y=_ptr(y), y_bstride=ybs, ldy=ldy)
    return y


def aa_activation(x: torch.Tensor, y: torch.Tensor, up_filter: torch.Tensor, down_filter: torch.Tensor, alpha: torch.Tensor, inv_beta: torch.Tensor, lens=None):
    """BigVGAN's anti-aliased SnakeBeta (``Activation1d``, codec/models/bigvgan/resample.py:157-177): x [B, L, C] -> y [B, L, C]."""
    B, L, C, xbs, ldx = _nlc(x)
    _, _, _, ybs, ldy = _nlc(y)
    assert up_filter.numel() == 12 and down_filter.numel() == 12 and alpha.numel() >= C and inv_beta.numel() >= C
    _lib.call_struct("mi355_aa_activation", "mi355_aa_act_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, C=C, L=L, lens=_ptr(lens), B=B,
                     up_filter=_ptr(up_filter), down_filter=_ptr(down_filter), alpha=_ptr(alpha), inv_beta=_ptr(inv_beta), y=_ptr(y), y_bstride=ybs, ldy=ldy)
    return y


def resample_poly(x: torch.Tensor, table: torch.Tensor, up: int, down: int, first: int, n_out: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Polyphase FIR rate conversion of the rows of x [rows, n_in] (``scipy.signal.resample_poly(..., padtype="edge")`` as ``mlx_audio/resample.py:40-47``
    calls it) with the tap-major float64 table [K, up] of ``mlx_audio_amd.resample.polyphase_table``."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and table.dtype == torch.float64 and table.is_contiguous() and table.shape[1] == up
    rows, n_in = x.shape
    if y is None:
        y = torch.empty(rows, n_out, device=x.device, dtype=torch.float32)
    assert y.shape == (rows, n_out) and y.stride(1) == 1 and y.dtype == torch.float32
    _lib.call_struct("mi355_resample_poly", "mi355_resample_args", _stream(), x=_ptr(x), x_bstride=x.stride(0), n_in=n_in, rows=rows, table=_ptr(table),
                     up=up, down=down, K=table.shape[0], first=first, y=_ptr(y), y_bstride=y.stride(0), n_out=n_out)
    return y


def conv1d_c1_k3s2(x: torch.Tensor, w3, bias: float, y: torch.Tensor, col: int, lens_in=None):
    """x [B, Lin] -> y[b, l, col] (Decoder.F0_conv / N_conv)."""
    assert x.dim() == 2 and x.stride(1) == 1
    B, Lout, _, ybs, ldy = _nlc(y)
    lib = _lib.load()
    rc = lib.mi355_conv1d_c1_k3s2(_ptr(x), x.stride(0), x.shape[1], _ptr(lens_in), float(w3[0]), float(w3[1]), float(w3[2]),
                                  float(bias), _ptr(y), ybs, ldy, col, Lout, B, _stream())
    _lib.check(rc, "mi355_conv1d_c1_k3s2")
    return y


# --------------------------------------------------------------------------------------- source / stft heads
def sine_source(f0: torch.Tensor, rand_ini: torch.Tensor, noise: torch.Tensor, lin_w: torch.Tensor, lin_b: float, up: int,
                lens2=None, sr: float = 24000.0, sine_amp: float = 0.1, noise_std: float = 0.003, voiced_thr: float = 10.0, quant: bool = False,
                coarse_f32: bool = False):
    B, L2 = f0.shape
    H = rand_ini.shape[1]
    assert f0.stride(1) == 1 and noise.is_contiguous() and tuple(noise.shape) == (B, L2 * up, H)
    ws = torch.empty((B, H, L2 + 1), dtype=torch.float32, device=f0.device)
    out = torch.zeros((B, L2 * up), dtype=torch.float32, device=f0.device)
    _lib.call_struct("mi355_sine_source", "mi355_sine_source_args", _stream(), f0=_ptr(f0), ld_f0=f0.stride(0), L2=L2,
                     lens2=_ptr(lens2), B=B, up=up, H=H, sr=sr, sine_amp=sine_amp, noise_std=noise_std, voiced_thr=voiced_thr,
                     rand_ini=_ptr(rand_ini), noise=_ptr(noise), lin_w=_ptr(lin_w), lin_b=lin_b, phase_ws=_ptr(ws),
                     out=_ptr(out), ld_out=out.stride(0),
                     quant_ws=_ptr(torch.empty((B, 2), dtype=torch.float32, device=f0.device)) if quant else None, coarse_f32=int(bool(coarse_f32)))
    return out


def stft_magphase(x: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, y: torch.Tensor, lens=None):
    assert x.dim() == 2 and x.stride(1) == 1
    B, _, _, ybs, ldy = _nlc(y)
    _lib.call_struct("mi355_stft_magphase", "mi355_stft_magphase_args", _stream(), x=_ptr(x), ldx=x.stride(0), L=x.shape[1],
                     lens=_ptr(lens), B=B, n_fft=n_fft, hop=hop, window=_ptr(window), y=_ptr(y), y_bstride=ybs, ldy=ldy)
    return y


def istft_head(x: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, audio: torch.Tensor, lens=None):
    B, Fr, _, xbs, ldx = _nlc(x)
    _lib.call_struct("mi355_istft_head", "mi355_istft_head_args", _stream(), x=_ptr(x), x_bstride=xbs, ldx=ldx, Fr=Fr,
                     lens=_ptr(lens), B=B, n_fft=n_fft, hop=hop, window=_ptr(window), audio=_ptr(audio),
                     ld_audio=audio.stride(0))
    return audio


# --------------------------------------------------------------------------------------- generic dsp
def stft_frames(x: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, pad_mode: int, n_frames: int):
    """x [B, L] -> complex64 [B, n_frames, n_fft//2+1]."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    B, L = x.shape
    out = torch.empty((B, n_frames, n_fft // 2 + 1, 2), dtype=torch.float32, device=x.device)
    _lib.call_struct("mi355_stft", "mi355_stft_args", _stream(), x=_ptr(x), ldx=x.stride(0), L=L, B=B, n_fft=n_fft, hop=hop,
                     window=_ptr(window), pad_mode=pad_mode, n_frames=n_frames, out=_ptr(out))
    return torch.view_as_complex(out)


def logmel(x: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, pad_mode: int, n_frames: int, fb: torch.Tensor,
           mode: int, log_guard: float = 0.0, fixed_max: Optional[float] = None):
    """Fused STFT -> power -> mel -> log (``mi355_logmel``).  ``mode`` 0 clamps to (max - 8) and rescales like Whisper: the maximum is the global one of
    every item, or ``fixed_max`` when given (Voxtral Realtime's ``global_log_mel_max``); ``mode`` 4 = ln(mel + ``log_guard``) (NeMo)."""
    assert x.dim() == 2 and x.stride(1) == 1 and fb.is_contiguous()
    B, L = x.shape
    n_mels = fb.shape[0]
    out = torch.empty((B, n_frames, n_mels), dtype=torch.float32, device=x.device)
    gmax = torch.empty((B,), dtype=torch.float32, device=x.device) if mode == 0 else None
    _lib.call_struct("mi355_logmel", "mi355_logmel_args", _stream(), x=_ptr(x), ldx=x.stride(0), L=L, B=B, n_fft=n_fft,
                     hop=hop, window=_ptr(window), pad_mode=pad_mode, n_frames=n_frames, fb=_ptr(fb), n_mels=n_mels,
                     mode=mode, out=_ptr(out), gmax=None if fixed_max is not None else _ptr(gmax), log_guard=float(log_guard))
    if mode == 0:
        if fixed_max is not None:
            gmax.fill_(float(fixed_max))
        lib = _lib.load()
        rc = lib.mi355_logmel_finish(_ptr(out), n_frames * n_mels, _ptr(gmax), B, _stream())
        _lib.check(rc, "mi355_logmel_finish")
    return out


def kaldi_frames(x: torch.Tensor, win: int, shift: int, pad: int, n_fft: int, n_frames: int, window: torch.Tensor, preemph: float,
                 noise: Optional[torch.Tensor] = None, dither: float = 0.0) -> torch.Tensor:
    """x [L] -> Kaldi frames [n_frames, n_fft] (DC removed, pre-emphasised, windowed, zero-padded)."""
    assert x.dim() == 1 and x.is_contiguous() and x.dtype == torch.float32 and window.numel() == win
    out = torch.empty((n_frames, n_fft), dtype=torch.float32, device=x.device)
    if noise is not None:
        assert noise.is_contiguous() and tuple(noise.shape) == (n_frames, win)
    _lib.call_struct("mi355_kaldi_frames", "mi355_kaldi_frames_args", _stream(), x=_ptr(x), L=x.numel(), win=win, shift=shift, pad=pad, n_fft=n_fft,
                     n_frames=n_frames, window=_ptr(window), noise=_ptr(noise), dither=dither, preemph=preemph, frames=_ptr(out))
    return out


def polar_spec(x: torch.Tensor, nb: int, clip: float = 1e2) -> torch.Tensor:
    """x [B, Fr, 2 nb] (log-magnitude | phase) -> complex64 [B, Fr, nb] = min(exp(m), clip) * exp(i p)  (Vocos ISTFTHead)."""
    B, Fr, C, xbs, ldx = _nlc(x)
    assert C >= 2 * nb
    spec = torch.empty((B, Fr, nb), dtype=torch.complex64, device=x.device)
    lib = _lib.load()
    rc = lib.mi355_polar_spec(_ptr(x), xbs, ldx, Fr, nb, B, float(clip), _ptr(torch.view_as_real(spec)), _stream())
    _lib.check(rc, "mi355_polar_spec")
    return spec


def istft_frames(spec: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, norm: torch.Tensor, norm_mode: int,
                 clamp: bool, trim: int, out_len: int):
    """spec complex64 [B, n_frames, nb] -> [B, out_len]."""
    sr = torch.view_as_real(spec.contiguous())
    B, n_frames = spec.shape[0], spec.shape[1]
    ws = torch.empty((B, (n_frames + 1) // 2 * 2, n_fft), dtype=torch.float32, device=spec.device)
    out = torch.empty((B, out_len), dtype=torch.float32, device=spec.device)
    _lib.call_struct("mi355_istft", "mi355_istft_args", _stream(), spec=_ptr(sr), n_frames=n_frames, B=B, n_fft=n_fft,
                     hop=hop, window=_ptr(window), norm=_ptr(norm), norm_mode=norm_mode, clamp=int(clamp), trim=trim,
                     out_len=out_len, frames_ws=_ptr(ws), out=_ptr(out), ld_out=out.stride(0))
    return out
