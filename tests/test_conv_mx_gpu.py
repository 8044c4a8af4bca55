"""conv precision 5 (fp16 hi pass + block-scaled e4m3 lo pass on v_mfma_scale_f32_32x32x64_f8f6f4) through the C ABI, against
 (a) the numpy statement of exactly that arithmetic (oracle/mx_ref.py): the kernel must perform the scheme it documents -- bar 3e-6 of the peak
     (fp32 accumulation order is the only freedom), and
 (b) the exact float64 conv: bar 3e-5 of the peak (the scheme's own error: ~2^-15 per element; one fp16 pass alone gives ~3e-4, bf16 hi + lo 3e-5).
Launches the wave-specialised kernel does not take run the precision-4 arithmetic on the image's fp16 slices (bar 3e-6 against float64)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mx_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from mlx_audio_amd import ops as _ops

    _ops.require_gpu()
    return _ops


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def peak_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def ref_conv_nlc(x, w, b, dil, pad):
    k = w.shape[1]
    xp = F.pad(x.transpose(1, 2).double(), (pad, (k - 1) * dil - pad))
    y = F.conv1d(xp, w.permute(0, 2, 1).double(), None if b is None else b.double(), dilation=dil)
    return y.transpose(1, 2)


def heavy_tailed(shape, g):
    a, b = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    return a * (1.0 + 3.0 * b * b)   # activations of a few units with heavier tails: the row maxima the block scales follow vary by 2^5


WS = 6128128
# the two MX images: 1 = e4m3 lo slices (precision 5), 2 = FP4 / e2m1 lo slices (precision 6).  Bars against the EXACT product follow the scheme's own
# error: ~2^-15 per element for e4m3 residuals (3e-5 of the peak), ~2^-13 for e2m1 residuals and weights (2e-4); the bar against the numpy statement of
# the scheme is the same 3e-6 for both -- the kernel must perform the arithmetic it documents
EXACT_BAR = {1: 3e-5, 2: 2e-4}
SCHEME = {1: mx_ref.conv_mx, 2: mx_ref.conv_mx4}


@pytest.mark.parametrize("cin,cout,k,dil,L,B,tile,on_ws4", [
    (128, 128, 7, 3, 300, 2, WS, True),
    (128, 128, 3, 1, 1500, 3, WS, True),
    (256, 256, 11, 5, 257, 1, WS, True),
    (128, 128, 11, 1, 129, 1, WS, True),
    (96, 200, 3, 1, 131, 2, WS, True),          # ragged channel counts: 3 chunks, 2 column tiles
    (1090, 300, 3, 1, 145, 1, WS, True),        # 35 chunks (odd): the two register-set roles end a tile swapped
    (64, 128, 3, 16, 260, 1, WS, True),         # window of 160 rows
    (128, 128, 3, 1, 70000, 1, WS, True),       # more row tiles than persistent workgroups
    (128, 384, 7, 1, 9000, 2, 0, True),         # the auto choice takes the wave-specialised kernel (>= 128 tiles)
    (128, 128, 7, 3, 300, 2, 0, False),         # few tiles: precision-4 arithmetic on the fp16 slices (4-wave kernels)
    (256, 256, 11, 5, 257, 1, 128128, False),
    (128, 128, 5, 1, 9000, 2, 0, False),        # K % 4 != 3: never the MX kernel
    (256, 128, 1, 1, 9000, 2, 0, False),
    (128, 22, 7, 1, 500, 1, 0, False),          # thin output
])
@pytest.mark.parametrize("mx", [1, 2])
def test_conv_precision5_plain(ops, cin, cout, k, dil, L, B, tile, on_ws4, mx):
    g = torch.Generator().manual_seed(cin + cout + k + dil)
    w = bf16r(torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    ld = ops.round_up(cin, 32)
    x = heavy_tailed((B, L, ld), g)
    pad = (k * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV, mx=mx)
    y = torch.full((B, L, ops.round_up(cout, 4)), float("nan"), device=DEV)
    ops.conv_gemm(x.to(DEV)[:, :, :cin], pc, y[:, :, :cout], dil=dil, pad=pad, tile=tile)
    torch.cuda.synchronize()
    got = y[:, :, :cout].cpu()
    assert torch.isfinite(got).all()
    exact = ref_conv_nlc(x[:, :, :cin], w, bias, dil, pad)
    e_exact = peak_err(got, exact)
    if on_ws4:
        nb = min(B, 2)
        scheme = np.stack([SCHEME[mx](x[b, :min(L, 3000), :cin].numpy(), w.numpy(), dil, pad) + bias.double().numpy() for b in range(nb)])
        rows = min(L, 3000) - (k - 1) * dil   # the oracle evaluated a prefix: its last rows lack their right context
        e_scheme = peak_err(got[:nb, :rows], scheme[:, :rows])
        print(f"precision {4 + mx} {cin}->{cout} k{k} d{dil}: vs scheme {e_scheme:.2e}, vs exact {e_exact:.2e}")
        assert e_scheme < 3e-6, e_scheme
        assert 1e-7 < e_exact < EXACT_BAR[mx], e_exact   # (the lower bound: the lo pass really is 8- / 4-bit -- fp16 hi + lo would sit at ~1e-7)
    else:
        assert e_exact < 3e-6, e_exact


@pytest.mark.parametrize("tile,res_shift,act", [(WS, 0, "snake"), (WS, 1, "snake"), (WS, 0, "leaky"), (WS, 0, "none"), (0, 0, "snake"), (2064128, 1, "snake")])
@pytest.mark.parametrize("mx", [1, 2])
def test_conv_precision5_fused_ragged(ops, tile, res_shift, act, mx):
    """AdaIN affine + Snake / LeakyReLU in front, bias + residual(row >> res_shift) + running sum + scale behind, ragged batch, fused statistics."""
    g = torch.Generator().manual_seed(11)
    B, L, C, K, dil = 3, 400, 128, 7, 3
    lens = torch.tensor([400, 64, 131], dtype=torch.int32)
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1
    x = heavy_tailed((B, L, C), g)
    sc, sh = torch.rand(B, C, generator=g) + 0.5, torch.randn(B, C, generator=g) * 0.3
    alpha = torch.rand(C, generator=g) + 0.5
    res, y0 = torch.randn(B, L, C, generator=g), torch.randn(B, L, C, generator=g)
    pad = (K * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV, mx=mx)
    lens_d = lens.to(DEV)
    y = y0.clone().to(DEV)
    kw = dict(pre_act=ops.ACT_SNAKE, pre_alpha=alpha.to(DEV)) if act == "snake" else (dict(pre_act=ops.ACT_LEAKY, pre_slope=0.2) if act == "leaky" else {})
    ops.conv_gemm(x.to(DEV), pc, y, dil=dil, pad=pad, lens_in=lens_d, lens_out=lens_d, pre=(sc.to(DEV), sh.to(DEV)), res=res.to(DEV),
                  res_shift=res_shift, out_scale=0.5, accumulate=True, tile=tile, **kw)
    torch.cuda.synchronize()
    got = y.cpu()
    for b in range(B):
        n = int(lens[b])
        t = x[b:b + 1, :n].double() * sc[b].double() + sh[b].double()
        if act == "snake":
            t = t + (1.0 / alpha.double()) * torch.sin(alpha.double() * t) ** 2
        elif act == "leaky":
            t = F.leaky_relu(t, 0.2)
        ref = (ref_conv_nlc(t, w, bias, dil, pad)[0] + res[b, torch.arange(n) >> res_shift].double() + y0[b, :n].double()) * 0.5
        assert peak_err(got[b, :n], ref) < EXACT_BAR[mx], (b, peak_err(got[b, :n], ref))
        assert torch.equal(got[b, n:], y0[b, n:])  # rows beyond the item's length are never written


@pytest.mark.parametrize("mx", [1, 2])
def test_conv_precision5_saturating_input(ops, mx):
    """|t| beyond fp16's range: the hi image saturates at 65504 and the residual carries the rest (coarsely: 3 significant bits) -- finite output,
    no NaN from an infinite half."""
    g = torch.Generator().manual_seed(3)
    C, K, L = 128, 3, 2000
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    x = torch.randn(1, L, C, generator=g)
    x[0, 100, 7] = 3.0e5
    x[0, 101, 9] = -7.0e4
    pc = ops.pack_conv(w, None, DEV, mx=mx)
    y = torch.empty(1, L, C, device=DEV)
    ops.conv_gemm(x.to(DEV), pc, y, dil=1, pad=1, tile=WS)
    torch.cuda.synchronize()
    got = y.cpu()
    assert torch.isfinite(got).all()
    ref = ref_conv_nlc(x, w, None, 1, 1)
    assert peak_err(got, ref) < (8e-2 if mx == 1 else 3e-1)   # the rows under the spikes: |lo| = |t| - 65504 carried with 3 (e4m3) / 1 (e2m1) mantissa bits
    far = torch.ones(L, dtype=torch.bool)
    far[96:106] = False
    assert peak_err(got[0, far], ref[0, far]) < EXACT_BAR[mx]
