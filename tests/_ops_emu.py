"""TEST INFRASTRUCTURE: a float64 CPU emulation of the CONTRACTS of a handful of ``mlx_audio_amd.ops`` entry points (``include/mi355audio.h``), used
only to dry-run HOST SCHEDULES on the CPU suite: the views, paddings, regrouped buffers, operand order and epilogue flags a model file hands to the
kernels are checked against the oracle before a GPU is spent on them.  It is not a fallback: nothing under ``mlx_audio_amd/`` imports it, the product
raises without the HIP library, and the parity claims rest on the ``-m gpu`` tests alone.  ``patched()`` swaps the entry points on the ``ops`` module
for the duration of a ``with`` block."""
from __future__ import annotations

import contextlib

import torch

from mlx_audio_amd import ops

ACT_NONE, ACT_LEAKY, ACT_SNAKE, ACT_GELU, ACT_ELU, ACT_SILU, ACT_GELU_TANH, ACT_TANH = 0, 1, 2, 3, 4, 5, 6, 7


def _act(v, act, slope=0.0):
    if act == ACT_NONE:
        return v
    if act == ACT_LEAKY:
        return torch.where(v > 0, v, v * slope)
    if act == ACT_ELU:
        return torch.where(v > 0, v, torch.expm1(v))
    if act == ACT_TANH:
        return torch.tanh(v)
    if act == ACT_GELU:
        return torch.nn.functional.gelu(v)
    if act == ACT_SILU:
        return torch.nn.functional.silu(v)
    if act == ACT_GELU_TANH:
        return torch.nn.functional.gelu(v, approximate="tanh")
    raise NotImplementedError(act)


def pack_conv(w, bias, device, f16=False, mx=False):
    if w.dim() == 2:
        w = w[:, None, :]
    w = w.detach().double().contiguous()
    cout, k, cin = w.shape
    return ops.PackedConv(w, None if bias is None else bias.detach().double(), cout, k, cin, f16, mx)


def pack_conv_transpose(w_t, bias, stride, device, f16=False):
    return pack_conv(ops._polyphase_weight(w_t.to(torch.float32), stride), bias, device, f16)


def pack_rowmajor16(w, bias, device, f16=False):
    n, k = w.shape
    return ops.RowMajor16(w.detach().double(), None if bias is None else bias.detach().double(), n, k, f16)


def pack_lstm_seq_wh(wh, device, f16=False):
    return pack_rowmajor16(wh, None, device, f16)   # block order: the emulated recurrence reads Wh as the reference does


def conv_gemm(x, pc, y, *, dil=1, pad=0, lens_in=None, lens_out=None, lout=None, pre=None, pre_act=ACT_NONE, pre_slope=0.0, pre_alpha=None,
              post_act=ACT_NONE, post_slope=0.0, res=None, res_shift=0, out_scale=1.0, accumulate=False, up=None, precision=2, tile=0, flat=None,
              use_bias=True, stats=None, pre_inv_beta=None, colscale=None, x_off=0, flatten=False, pre_fq=None):
    """mi355_conv_gemm_args, restated: y = epilogue(conv1d(prologue(x))) on channels-last rows (see the header for every field)."""
    assert stats is None and pre_fq is None, "not emulated"
    B, Lx = x.shape[0], x.shape[1]
    Lout = lout if lout is not None else y.shape[1]
    K, Cin, Cout = pc.k, pc.cin, pc.cout
    w = pc.w  # [Cout, K, Cin] float64
    rows = torch.arange(Lout)
    taps = []
    for b in range(B):
        len_in = int(lens_in[b]) if lens_in is not None else Lx
        if flat is not None:   # row r, tap k, channel c reads flat element (r - pad + k * dil) * ldx + x_off + c, valid inside [0, len_in * channels)
            xf = x[b].reshape(-1).double()
            idx = ((rows[:, None, None] - pad + torch.arange(K)[None, :, None] * dil) * flat["ldx"] + flat["x_off"] + torch.arange(Cin)[None, None, :])
            ok = (idx >= 0) & (idx < len_in * flat["channels"])
            t = torch.where(ok, xf[idx.clamp(0, xf.numel() - 1)], torch.zeros((), dtype=torch.float64))
        else:
            xr = x[b, :, :Cin].double() if x_off == 0 else x[b].reshape(-1)[x_off:].double().reshape(Lx, -1)[:, :Cin]
            idx = rows[:, None] - pad + torch.arange(K)[None, :] * dil       # [Lout, K]
            ok = (idx >= 0) & (idx < len_in)
            t = xr[idx.clamp(0, Lx - 1)]                                      # [Lout, K, Cin]
        # prologue on the gathered values (per input element in the kernel: the same numbers), padding rows are zero AFTER it
        if pre is not None:
            sc, sh = pre
            t = t * sc[b, :Cin].double() + sh[b, :Cin].double()
        if pre_act == ACT_SNAKE:
            al = pre_alpha[:Cin].double()
            inv = pre_inv_beta[:Cin].double() if pre_inv_beta is not None else 1.0 / al
            t = t + inv * torch.sin(al * t) ** 2
        else:
            t = _act(t, pre_act, pre_slope)
        t = torch.where(ok if flat is not None else ok[:, :, None], t, torch.zeros((), dtype=torch.float64))
        taps.append(t)
    t = torch.stack(taps)                                                     # [B, Lout, K, Cin]
    acc = torch.einsum("blkc,nkc->bln", t, w)
    if use_bias and pc.bias is not None:   # polyphase store: GEMM column r * up_cout + co carries output channel co's bias
        acc = acc + (pc.bias if up is None else pc.bias.repeat(up["s"]))
    v = _act(acc, post_act, post_slope)
    if colscale is not None:
        v = v * colscale[:Cout].double()
    if up is not None:
        assert res is None and not accumulate
        s, p, co, Lup, off = up["s"], up["p"], up["cout"], up["lout"], up.get("row_off", 0)
        for r in range(s):
            tgt = rows * s + r - p + off
            keep = (tgt >= 0) & (tgt < Lup)
            y[:, tgt[keep], :co] = (v[:, keep, r * co:(r + 1) * co] * out_scale).to(y.dtype)
        return y
    if res is not None:
        v = v + res[:, (rows >> res_shift), :Cout].double()
    if accumulate:
        v = v + y[:, :Lout, :Cout].double()
    v = v * out_scale
    if lens_out is not None:
        for b in range(B):
            y[b, :int(lens_out[b]), :Cout] = v[b, :int(lens_out[b])].to(y.dtype)
    else:
        y[:, :Lout, :Cout] = v.to(y.dtype)
    return y


def embed_sum(table, ids, y, *, slot_offset=None, add=None, scale=1.0, lens=None):
    B, L, Q = ids.shape
    acc = add.double().clone() if add is not None else torch.zeros(y.shape, dtype=torch.float64)
    for q in range(Q):
        off = int(slot_offset[q]) if slot_offset is not None else 0
        i = ids[:, :, q].long()
        acc = acc + torch.where((i >= 0)[:, :, None], table[(i.clamp(min=0) + off)].double(), torch.zeros((), dtype=torch.float64))
    y.copy_((acc * scale).to(y.dtype))
    return y


def rvq_encode(x, tables, tables_t, c2, margins=False):
    rows, D = x.shape
    n, bins, _ = tables.shape
    assert tuple(tables_t.shape) == (n, D, bins) and torch.equal(tables_t, tables.transpose(1, 2))
    r = x.float().clone()
    codes = torch.empty((rows, n), dtype=torch.int32)
    mg = torch.empty((rows, n), dtype=torch.float32)
    for l in range(n):
        s = c2[l][None, :] - r @ tables_t[l]
        top = torch.topk(-s, 2, dim=1).values
        idx = s.argmin(1)
        codes[:, l] = idx.to(torch.int32)
        mg[:, l] = top[:, 0] - top[:, 1]
        r = r - tables[l][idx]
    return (codes, mg) if margins else codes


def dwconv(x, w, bias, y, *, pad=0, stride=1, transpose=False, lens_in=None, dil=1, pre_alpha=None, pre_inv=None):
    assert lens_in is None
    B, Lin, C = x.shape
    Lout, K = y.shape[1], w.shape[1]
    xd = x.double()
    if pre_alpha is not None:
        xd = xd + pre_inv[:C].double() * torch.sin(pre_alpha[:C].double() * xd) ** 2
    if not transpose:
        rows = torch.arange(Lout)[:, None] * stride + torch.arange(K)[None, :] * dil - pad
        ok = (rows >= 0) & (rows < Lin)
        t = torch.where(ok[None, :, :, None], xd[:, rows.clamp(0, Lin - 1)], torch.zeros((), dtype=torch.float64))   # [B, Lout, K, C]
        v = torch.einsum("blkc,ck->blc", t, w.double())
    else:
        v = torch.zeros((B, Lout, C), dtype=torch.float64)
        for k in range(K):
            tgt = torch.arange(Lin) * stride + k - pad
            keep = (tgt >= 0) & (tgt < Lout)
            v[:, tgt[keep]] += xd[:, keep] * w[:, k].double()
    if bias is not None:
        v = v + bias.double()
    y.copy_(v.to(y.dtype))
    return y


def lstm_seq(xproj, wh, out, h0=None, c0=None):
    B, T, H4 = xproj.shape
    H = H4 // 4
    h = torch.zeros((B, H), dtype=torch.float64) if h0 is None else h0.double().clone()
    c = torch.zeros((B, H), dtype=torch.float64) if c0 is None else c0.double().clone()
    for t in range(T):
        g = xproj[:, t].double() + h @ wh.w.t()
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        out[:, t] = h.to(out.dtype)
    return h.float(), c.float()


@contextlib.contextmanager
def patched():
    names = dict(require_gpu=lambda: None, pack_conv=pack_conv, pack_conv_transpose=pack_conv_transpose, pack_rowmajor16=pack_rowmajor16, pack_lstm_seq_wh=pack_lstm_seq_wh, conv_gemm=conv_gemm,
                 embed_sum=embed_sum, rvq_encode=rvq_encode, dwconv=dwconv, lstm_seq=lstm_seq)
    saved = {k: getattr(ops, k) for k in names}
    try:
        for k, v in names.items():
            setattr(ops, k, v)
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
