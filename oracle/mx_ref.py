"""numpy statement of the arithmetic of conv precision 5 and of its weight image (TEST ORACLE, not product).

Precision 5 (include/mi355audio.h, ``mi355_conv_gemm_args.precision``) is this repository's own number format for the vocoder convs -- the
reference (``/root/reference/mlx_audio/tts/models/kokoro/istftnet.py:128-170`` ``ConvWeighted``) computes ``mx.conv1d`` in the checkpoint dtype
and knows nothing of it -- so what is pinned here is (a) that the HIP kernel performs exactly the arithmetic it documents and (b) how far that
arithmetic is from the exact product:

    y = sum fp16(t) * w                                   (fp16 hi pass; bf16-valued weights are exact in fp16)
      + sum q8(t - fp16(t)) * q8(w)                       (OCP MX: e4m3 elements, one E8M0 scale per window row and 32-channel chunk for the
                                                           activations, one per output column for the weights)

``t`` = the conv input AFTER its prologue.  OCP e4m3fn: 4 exponent bits (bias 7), 3 mantissa bits, subnormal quantum 2^-9, max finite 448,
round to nearest even; E8M0: scale = 2^(byte - 127).
"""
from __future__ import annotations

import numpy as np


def e4m3_round(v: np.ndarray) -> np.ndarray:
    """Nearest e4m3fn value (ties to even), saturating at +-448; float64 in, float64 out."""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)                      # below the smallest normal 2^-6 the grid is the subnormal one: quantum 2^-9
    q = np.exp2(e - 3.0)
    r = np.rint(a / q) * q                       # np.rint: ties to even
    r = np.minimum(r, 448.0)
    return np.sign(v) * r


def e4m3_bits(v: np.ndarray) -> np.ndarray:
    """e4m3fn code (uint8) of values that already lie on the grid."""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    s = (np.signbit(v)).astype(np.uint8) << 7
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    sub = a < 2.0 ** -6
    ee = np.where(sub, 0, e + 7).astype(np.int64)
    m = np.where(sub, a / 2.0 ** -9, (a / np.exp2(e) - 1.0) * 8.0)
    m = np.rint(m).astype(np.int64)
    assert ((m >= 0) & (m <= 7)).all()
    return (s | (ee.astype(np.uint8) << 3) | m.astype(np.uint8)).astype(np.uint8)


def column_scale_exponents(w: np.ndarray) -> np.ndarray:
    """E8M0 exponent e[n] of every output column of w [Cout, K, Cin]: floor(log2(max |w[n]|)) - 7 (0 for an all-zero column)."""
    amax = np.abs(w.reshape(w.shape[0], -1)).max(axis=1).astype(np.float64)
    e = np.zeros(w.shape[0], dtype=np.int64)
    nz = amax > 0
    e[nz] = np.floor(np.log2(amax[nz])).astype(np.int64) - 7
    return np.clip(e, -127, 120)


def quantise_weights(w: np.ndarray) -> np.ndarray:
    """The values the lo pass multiplies by: e4m3(w / 2^e[n]) * 2^e[n]."""
    e = column_scale_exponents(w)
    sc = np.exp2(e.astype(np.float64))[:, None, None]
    return e4m3_round(w.astype(np.float64) / sc) * sc


def split_activation(t: np.ndarray):
    """t [..., C] float32 (C a multiple of 32, zero padded) -> (hi, lo_q) float64: hi = fp16(clamp(t)), lo_q = the MX-quantised residual with one
    shared exponent per row and 32-channel chunk: floor(log2(max |lo|)) - 7, floored at byte 1 = 2^-126 (the kernel hands the scale to
    v_cvt_scalef32_pk_fp8_f32 as a float, which must be a normal number; a row that small converts to zeros anyway)."""
    t = np.asarray(t, dtype=np.float32)
    hi = np.clip(t, -65504.0, 65504.0).astype(np.float16).astype(np.float32)
    lo = (t - hi).astype(np.float32)                        # exact in float32
    C = t.shape[-1]
    assert C % 32 == 0
    blk = lo.reshape(*t.shape[:-1], C // 32, 32)
    amax = np.abs(blk).max(axis=-1)
    ef = (amax.view(np.uint32) >> 23).astype(np.int64)      # biased float32 exponent of the block maximum
    sb = np.maximum(ef - 7, 1)
    scale = np.exp2((sb - 127).astype(np.float64))[..., None]
    q = e4m3_round(blk.astype(np.float64) / scale) * scale
    return hi.astype(np.float64), q.reshape(t.shape)


def conv_mx(t: np.ndarray, w: np.ndarray, dil: int, pad: int) -> np.ndarray:
    """t [L, Cin] float32 (prologue output), w [Cout, K, Cin] (bf16-valued) -> [L, Cout] float64 by the precision-5 arithmetic, 'same' length."""
    L, cin = t.shape
    cout, K, _ = w.shape
    cp = (cin + 31) // 32 * 32
    tp = np.zeros((L, cp), dtype=np.float32)
    tp[:, :cin] = t
    hi, lo = split_activation(tp)
    w16 = w.astype(np.float16).astype(np.float64)
    wq = quantise_weights(w)
    y = np.zeros((L, cout), dtype=np.float64)
    for k in range(K):
        off = k * dil - pad
        lo_r, hi_r = max(0, -off), min(L, L - off)
        if hi_r <= lo_r:
            continue
        rows = slice(lo_r + off, hi_r + off)
        y[lo_r:hi_r] += hi[rows, :cin] @ w16[:, k, :].T + lo[rows, :cin] @ wq[:, k, :].T
    return y


def pack_mx_image(w: np.ndarray) -> np.ndarray:
    """Independent restatement of ``mi355_pack_conv_weight_mx_host`` (csrc/api.cpp): uint8 image of w [Cout, K, Cin] float32."""
    cout, K, cin = w.shape
    chunks, ntp, npair = (cin + 31) // 32, (cout + 127) // 128 * 4, (K + 1) // 2
    e = column_scale_exponents(w)
    wp = np.zeros((ntp * 32, 2 * npair, chunks * 32), dtype=np.float64)
    wp[:cout, :K, :cin] = w
    ep = np.zeros(ntp * 32, dtype=np.int64)
    ep[:cout] = e
    codes = e4m3_bits(e4m3_round(wp / np.exp2(ep.astype(np.float64))[:, None, None]))      # [N, 2 NP, C]
    h16 = wp[:, :K, :].astype(np.float16).view(np.uint16)                                   # [N, K, C]
    out = np.zeros(chunks * (K + npair) * ntp * 2048 + ntp * 32, dtype=np.uint8)
    lane = np.arange(64)
    n_of = lane & 31
    for ch in range(chunks):
        base = ch * (K + npair) * ntp * 2048
        for tap in range(K):
            for nt in range(ntp):
                for kk in range(2):
                    c0 = ch * 32 + kk * 16 + (lane >> 5) * 8
                    frag = h16[nt * 32 + n_of][:, tap][np.arange(64)[:, None], c0[:, None] + np.arange(8)[None, :]]       # [64, 8]
                    o = base + ((tap * ntp + nt) * 2 + kk) * 1024
                    out[o:o + 1024] = frag.astype(np.uint16).reshape(-1).view(np.uint8)
        for p in range(npair):
            for nt in range(ntp):
                for h in range(2):
                    c0 = ch * 32 + (lane >> 5) * 16
                    frag = codes[nt * 32 + n_of][:, 2 * p + h][np.arange(64)[:, None], c0[:, None] + np.arange(16)[None, :]]  # [64, 16]
                    o = base + K * ntp * 2048 + ((p * ntp + nt) * 2 + h) * 1024
                    out[o:o + 1024] = frag.reshape(-1)
    out[chunks * (K + npair) * ntp * 2048:] = (ep + 127).astype(np.uint8)
    return out


# ------------------------------------------------------------------------------------------------ precision 6: FP4 (OCP e2m1) lo pass
# y = sum fp16(t) * w + sum q4(t - fp16(t)) * q4(w): e2m1 elements {0, 0.5, 1, 1.5, 2, 3, 4, 6} (sign + 2 exponent bits, bias 1, + 1 mantissa bit), one E8M0
# scale per window row and 32-channel chunk for the activations and one per output column for the weights, both chosen as floor(log2(max)) - 2 (OCP MX: the
# scaled maximum lies in [4, 8), values above 6 saturate); round to nearest, ties to the even CODE (what v_cvt_scalef32_pk_fp4_f32 does: probed on gfx950,
# profiles/r6_mfma_fp4_probe_call2.jsonl).
_E2M1 = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def e2m1_code(v: np.ndarray) -> np.ndarray:
    """e2m1 code (0..15, bit 3 = sign) of the nearest grid value (ties to the even code), saturating at +-6; float64 in."""
    v = np.asarray(v, dtype=np.float64)
    a = np.minimum(np.abs(v), 6.0)
    d = np.abs(a[..., None] - _E2M1)
    best = d.min(axis=-1, keepdims=True)
    cand = d == best
    lo = cand.argmax(axis=-1)                                   # the lower candidate
    hi = 7 - cand[..., ::-1].argmax(axis=-1)                    # the upper candidate (== lo unless a tie)
    code = np.where(lo == hi, lo, np.where(lo % 2 == 0, lo, hi))
    return (code | (np.signbit(v).astype(np.int64) << 3)).astype(np.uint8)


def e2m1_round(v: np.ndarray) -> np.ndarray:
    c = e2m1_code(v)
    return np.where(c & 8, -1.0, 1.0) * _E2M1[c & 7]


def column_scale_exponents4(w: np.ndarray) -> np.ndarray:
    """E8M0 exponent e[n] of every output column of w [Cout, K, Cin] for the FP4 image: floor(log2(max |w[n]|)) - 2 (0 for an all-zero column)."""
    amax = np.abs(w.reshape(w.shape[0], -1)).max(axis=1).astype(np.float64)
    e = np.zeros(w.shape[0], dtype=np.int64)
    nz = amax > 0
    e[nz] = np.floor(np.log2(amax[nz])).astype(np.int64) - 2
    return np.clip(e, -127, 120)


def quantise_weights4(w: np.ndarray) -> np.ndarray:
    e = column_scale_exponents4(w)
    sc = np.exp2(e.astype(np.float64))[:, None, None]
    return e2m1_round(w.astype(np.float64) / sc) * sc


def split_activation4(t: np.ndarray):
    """As split_activation with FP4 residuals: shared exponent per row and 32-channel chunk = (biased float32 exponent of the block maximum) - 2, floored at byte 1."""
    t = np.asarray(t, dtype=np.float32)
    hi = np.clip(t, -65504.0, 65504.0).astype(np.float16).astype(np.float32)
    lo = (t - hi).astype(np.float32)
    C = t.shape[-1]
    assert C % 32 == 0
    blk = lo.reshape(*t.shape[:-1], C // 32, 32)
    amax = np.abs(blk).max(axis=-1)
    ef = (amax.view(np.uint32) >> 23).astype(np.int64)
    sb = np.maximum(ef - 2, 1)
    scale = np.exp2((sb - 127).astype(np.float64))[..., None]
    q = e2m1_round(blk.astype(np.float64) / scale) * scale
    return hi.astype(np.float64), q.reshape(t.shape)


def conv_mx4(t: np.ndarray, w: np.ndarray, dil: int, pad: int) -> np.ndarray:
    """conv_mx by the precision-6 arithmetic."""
    L, cin = t.shape
    cout, K, _ = w.shape
    cp = (cin + 31) // 32 * 32
    tp = np.zeros((L, cp), dtype=np.float32)
    tp[:, :cin] = t
    hi, lo = split_activation4(tp)
    w16 = w.astype(np.float16).astype(np.float64)
    wq = quantise_weights4(w)
    y = np.zeros((L, cout), dtype=np.float64)
    for k in range(K):
        off = k * dil - pad
        lo_r, hi_r = max(0, -off), min(L, L - off)
        if hi_r <= lo_r:
            continue
        rows = slice(lo_r + off, hi_r + off)
        y[lo_r:hi_r] += hi[rows, :cin] @ w16[:, k, :].T + lo[rows, :cin] @ wq[:, k, :].T
    return y


def pack_mx4_image(w: np.ndarray) -> np.ndarray:
    """Independent restatement of ``mi355_pack_conv_weight_mx4_host`` (csrc/api.cpp): the MX image's geometry, one kilobyte of e2m1 codes per 32-column
    group and tap pair (lane l: column l & 31, tap 2 p + (l >> 5), nibble j = channel j of the chunk, low nibble first), the second kilobyte zero."""
    cout, K, cin = w.shape
    chunks, ntp, npair = (cin + 31) // 32, (cout + 127) // 128 * 4, (K + 1) // 2
    e = column_scale_exponents4(w)
    wp = np.zeros((ntp * 32, 2 * npair, chunks * 32), dtype=np.float64)
    wp[:cout, :K, :cin] = w
    ep = np.zeros(ntp * 32, dtype=np.int64)
    ep[:cout] = e
    codes = e2m1_code(wp / np.exp2(ep.astype(np.float64))[:, None, None])                   # [N, 2 NP, C]
    codes[cout:] = 0
    codes[:, K:] = 0
    codes[:, :, cin:] = 0
    h16 = wp[:, :K, :].astype(np.float16).view(np.uint16)
    out = np.zeros(chunks * (K + npair) * ntp * 2048 + ntp * 32, dtype=np.uint8)
    lane = np.arange(64)
    n_of = lane & 31
    for ch in range(chunks):
        base = ch * (K + npair) * ntp * 2048
        for tap in range(K):
            for nt in range(ntp):
                for kk in range(2):
                    c0 = ch * 32 + kk * 16 + (lane >> 5) * 8
                    frag = h16[nt * 32 + n_of][:, tap][np.arange(64)[:, None], c0[:, None] + np.arange(8)[None, :]]
                    o = base + ((tap * ntp + nt) * 2 + kk) * 1024
                    out[o:o + 1024] = frag.astype(np.uint16).reshape(-1).view(np.uint8)
        for p in range(npair):
            for nt in range(ntp):
                c = codes[nt * 32 + n_of, 2 * p + (lane >> 5)][:, ch * 32:ch * 32 + 32]     # [64 lanes, 32 channels]
                o = base + K * ntp * 2048 + (p * ntp + nt) * 2048
                out[o:o + 1024] = (c[:, 0::2] | (c[:, 1::2] << 4)).astype(np.uint8).reshape(-1)
    out[chunks * (K + npair) * ntp * 2048:] = (ep + 127).astype(np.uint8)
    return out


# ---- SPLIT words (include/mi355audio.h: mi355_conv_gemm_args.x_split / y_split, mi355_split16) --------------------------------------------------
def bf16_round_bits(v: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bits (uint16), round to nearest even (finite inputs)."""
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def split16_words(v: np.ndarray, fmt: int) -> np.ndarray:
    """The uint32 SPLIT word of every float32 value: bits 0-15 = the 16-bit hi part, bits 16-31 = the 16-bit lo residual, in IEEE half (``fmt`` 4: the value
    clamped to +-65504 first) or bfloat16 (``fmt`` 2) -- the two numbers the conv prologue of that precision makes of the value (conv_ws4.h convertA)."""
    v = np.asarray(v, dtype=np.float32)
    if fmt == 4:
        c = np.clip(v, -65504.0, 65504.0).astype(np.float32)
        h = c.astype(np.float16)
        lo = (c - h.astype(np.float32)).astype(np.float32).astype(np.float16)   # the fp32 difference is exact: one rounding, to half
        return h.view(np.uint16).astype(np.uint32) | (lo.view(np.uint16).astype(np.uint32) << 16)
    assert fmt == 2
    hb = bf16_round_bits(v)
    hf = (hb.astype(np.uint32) << 16).view(np.float32)
    lb = bf16_round_bits((v - hf).astype(np.float32))
    return hb.astype(np.uint32) | (lb.astype(np.uint32) << 16)


def split16_value(words: np.ndarray, fmt: int) -> np.ndarray:
    """hi + lo of SPLIT words, as float64."""
    w = np.asarray(words, dtype=np.uint32)
    h16, l16 = (w & 0xffff).astype(np.uint16), (w >> 16).astype(np.uint16)
    if fmt == 4:
        return h16.view(np.float16).astype(np.float64) + l16.view(np.float16).astype(np.float64)
    return (h16.astype(np.uint32) << 16).view(np.float32).astype(np.float64) + (l16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
