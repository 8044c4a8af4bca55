"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle / fp64 torch.

All tests here need a real MI355X: run with ``pytest -m gpu``.  Tolerances are stated per test:
bit-exact for integer/index outputs, fp32-level for everything computed in fp32, and ~2^-16
relative for the bf16 hi+lo split GEMM (precision=2; precision=1 is the plain bf16 pass).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from mlx_audio_amd import ops as _ops

    _ops.require_gpu()
    return _ops


DEV = "cuda"


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def ref_conv_nlc(x, w, b, dil, pad):
    """x [B, L, Cin] fp64, w [Cout, K, Cin] -> [B, Lout, Cout] (zero padding `pad` left, symmetric right)."""
    k = w.shape[1]
    xp = F.pad(x.transpose(1, 2).double(), (pad, (k - 1) * dil - pad))  # output length == input length, also for even K
    y = F.conv1d(xp, w.permute(0, 2, 1).double(), None if b is None else b.double(), dilation=dil)
    return y.transpose(1, 2)


@pytest.mark.parametrize("cin,cout,k,dil,L,B,tile", [
    (128, 128, 7, 3, 300, 2, 0),
    (128, 128, 3, 1, 700, 1, 128128),
    (256, 256, 11, 5, 257, 1, 64128),
    (256, 256, 11, 5, 257, 1, 128128),
    (512, 64, 1, 1, 77, 3, 0),
    (1090, 1024, 3, 1, 45, 1, 0),
    (514, 200, 3, 1, 33, 2, 64064),
    (768, 50, 1, 1, 80, 1, 0),
    (128, 22, 7, 1, 500, 1, 0),
    # the wave-specialised producer / consumer kernel (tile code 6128128; +10000000 = consumers at default priority, +80000000 = one
    # workgroup per tile instead of persistent workgroups): conv mode with 1, 2 and 3 N-tiles, K = 1 / 2 / odd, dilation, ragged channel
    # counts, more row tiles than XCDs ...
    (128, 128, 7, 3, 300, 2, 6128128),
    (128, 128, 3, 1, 1500, 3, 6128128),
    (256, 256, 11, 5, 257, 1, 6128128),
    (128, 128, 11, 1, 129, 1, 6128128),
    (96, 200, 2, 1, 131, 2, 6128128),
    (1090, 300, 3, 1, 145, 1, 6128128),
    (32, 24, 1, 1, 500, 1, 6128128),
    (64, 128, 5, 16, 260, 1, 6128128),
    (128, 128, 4, 2, 2100, 2, 16128128),
    (128, 128, 7, 3, 300, 2, 86128128),
    (256, 256, 11, 5, 1200, 3, 86128128),
    (128, 128, 3, 1, 70000, 1, 6128128),
    # ... and GEMM mode (K = 1: 64-channel super-chunks; odd / ragged chunk counts, one row tile, many row tiles)
    (512, 64, 1, 1, 777, 3, 6128128),
    (160, 128, 1, 1, 2100, 1, 6128128),
    (768, 2304, 1, 1, 80, 3, 6128128),
    (1090, 1024, 1, 1, 264, 2, 6128128),
    (2048, 768, 1, 1, 5000, 1, 16128128),
    (64, 128, 1, 1, 130, 1, 6128128),
    # split-K (tile codes 2064128 / 2064064 + 10000000 * groups; tile 0 with few tiles and a deep K loop takes it by itself): the launches of a
    # single utterance -- PL-BERT projections, frame-level convs, the first generator stage
    (768, 768, 1, 1, 80, 1, 2064128),
    (3072, 768, 1, 1, 80, 1, 2064128),
    (768, 2304, 1, 1, 80, 1, 52064128),
    (256, 256, 11, 5, 5281, 1, 2064128),
    (256, 256, 11, 5, 257, 1, 42064128),
    (1090, 1024, 3, 1, 264, 1, 2064128),
    (514, 200, 3, 1, 33, 2, 22064128),
    (512, 64, 1, 1, 77, 3, 2064064),
    (96, 50, 7, 2, 130, 1, 32064064),
    # the wave-specialised kernel on 128 x 64 tiles (tile code 6128064; tile 0 takes it for C_out <= 64 once there are >= 128 row tiles):
    # conv mode, GEMM mode, ragged channel counts, one and many row tiles
    (64, 64, 7, 3, 300, 2, 6128064),
    (64, 64, 11, 5, 2100, 2, 6128064),
    (128, 64, 3, 1, 1500, 1, 6128064),
    (64, 50, 11, 1, 129, 1, 6128064),
    (256, 22, 7, 1, 500, 1, 6128064),
    (512, 64, 1, 1, 777, 3, 6128064),
    (64, 64, 1, 1, 130, 1, 6128064),
    (160, 40, 1, 1, 2100, 1, 16128064),
    (64, 64, 3, 1, 20000, 1, 0),
    (64, 64, 3, 1, 70000, 1, 86128064),
])
def test_conv_gemm_plain(ops, cin, cout, k, dil, L, B, tile):
    g = torch.Generator().manual_seed(cin + cout + k)
    w = bf16r(torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    ld = ops.round_up(cin, 32)
    x = torch.randn(B, L, ld, generator=g)
    pad = (k * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV)
    xd = x.to(DEV)
    y = torch.full((B, L, ops.round_up(cout, 4)), float("nan"), device=DEV)
    pc16 = ops.pack_conv(w, bias, DEV, f16=True)
    ref = ref_conv_nlc(x[:, :, :cin], w, bias, dil, pad)
    # 2: bf16 hi+lo split (~2^-16), 1: single bf16 pass (~2^-8), 3: single fp16 pass (~2^-11; fp16-packed weights),
    # 4: fp16 hi+lo split (~2^-22)
    for prec, tol, p in ((2, 3e-5, pc), (1, 1.5e-2, pc), (3, 2e-3, pc16), (4, 3e-6, pc16)):
        y.fill_(float("nan"))
        ops.conv_gemm(xd[:, :, :cin], p, y[:, :, :cout], dil=dil, pad=pad, precision=prec, tile=tile)
        torch.cuda.synchronize()
        got = y[:, :, :cout].cpu()
        assert torch.isfinite(got).all()
        assert rel_err(got, ref) < tol, (prec, rel_err(got, ref))


@pytest.mark.parametrize("tile", [0, 6128128, 2064128])
def test_conv_gemm_accumulate_then_out_scale(ops, tile):
    """The epilogue order the header states: y = (act(acc + bias) + res + old y) * out_scale -- out_scale multiplies the accumulated value too (the
    resblock-stage mean of iSTFTNet / BigVGAN: blocks j > 0 accumulate, the last one carries out_scale = 1 / num_kernels)."""
    g = torch.Generator().manual_seed(5)
    cin, cout, k, L, B = 128, 128, 3, 300, 2
    w = bf16r(torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    x, res, old = torch.randn(B, L, cin, generator=g), torch.randn(B, L, cout, generator=g), torch.randn(B, L, cout, generator=g)
    ref = (ref_conv_nlc(x, w, bias, 1, 1) + res.double() + old.double()) / 3.0
    y = old.to(DEV).clone()
    ops.conv_gemm(x.to(DEV), ops.pack_conv(w, bias, DEV), y, dil=1, pad=1, res=res.to(DEV), accumulate=True, out_scale=1.0 / 3.0, tile=tile)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref) < 3e-5, rel_err(y.cpu(), ref)


@pytest.mark.parametrize("tile,res_shift,act", [(0, 1, "snake"), (6128128, 1, "snake"), (6128128, 0, "snake"), (128128, 0, "leaky"),
                                                 (6128128, 0, "leaky"), (64064, 0, "snake"), (16128128, 0, "snake"),
                                                 (86128128, 1, "snake"), (86128128, 0, "snake"), (86128128, 0, "leaky"),
                                                 (2064128, 1, "snake"), (32064128, 0, "leaky"), (2064064, 0, "snake")])
def test_conv_gemm_fused_prologue_epilogue_ragged(ops, tile, res_shift, act):
    """AdaIN affine + Snake / LeakyReLU in front, bias + residual(row >> res_shift) + scale + accumulate behind, ragged
    batch; res_shift == 0 takes the accumulator-initialisation ("fold") path of the wave-specialised kernel."""
    g = torch.Generator().manual_seed(7)
    B, L, C, K, dil = 3, 210, 128, 7, 3
    lens = torch.tensor([210, 64, 131], dtype=torch.int32)
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, L, C, generator=g)
    sc = torch.rand(B, C, generator=g) + 0.5
    sh = torch.randn(B, C, generator=g) * 0.3
    alpha = torch.rand(C, generator=g) + 0.5
    res = torch.randn(B, L, C, generator=g)
    y0 = torch.randn(B, L, C, generator=g)
    pad = (K * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV)
    lens_d = lens.to(DEV)
    y = y0.clone().to(DEV)
    kw = dict(pre_act=ops.ACT_SNAKE, pre_alpha=alpha.to(DEV)) if act == "snake" else dict(pre_act=ops.ACT_LEAKY, pre_slope=0.2)
    ops.conv_gemm(x.to(DEV), pc, y, dil=dil, pad=pad, lens_in=lens_d, lens_out=lens_d, pre=(sc.to(DEV), sh.to(DEV)),
                  res=res.to(DEV), res_shift=res_shift, out_scale=0.5, accumulate=True, tile=tile, **kw)
    torch.cuda.synchronize()
    got = y.cpu()
    for b in range(B):
        n = int(lens[b])
        t = x[b:b + 1, :n].double() * sc[b].double() + sh[b].double()
        if act == "snake":
            t = t + (1.0 / alpha.double()) * torch.sin(alpha.double() * t) ** 2
        else:
            t = F.leaky_relu(t, 0.2)
        ref = ref_conv_nlc(t, w, bias, dil, pad)[0]
        ref = (ref + res[b, torch.arange(n) >> res_shift].double() + y0[b, :n].double()) * 0.5
        assert rel_err(got[b, :n], ref) < 3e-5
        assert torch.equal(got[b, n:], y0[b, n:])  # rows beyond the item's length are never written


@pytest.mark.parametrize("tile", [0, 6128064, 86128064, 64064])
def test_conv_gemm_thin_tiles_fused(ops, tile):
    """C = 64 (KittenTTS's last generator stage): AdaIN + Snake prologue, bias + residual + out_scale, fused instance-norm statistics, ragged batch,
    through the 128 x 64 tiles of the wave-specialised kernel (6128064) and through the 4-wave kernel it replaces (64064; no fused statistics there)."""
    g = torch.Generator().manual_seed(31)
    B, L, C, K, dil = 3, 1900, 64, 7, 3
    lens = torch.tensor([1900, 64, 1031], dtype=torch.int32)
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1 + 3.0
    x = torch.randn(B, L, C, generator=g)
    sc = torch.rand(B, C, generator=g) + 0.5
    sh = torch.randn(B, C, generator=g) * 0.3
    alpha = torch.rand(C, generator=g) + 0.5
    res = torch.randn(B, L, C, generator=g)
    gb = (torch.randn(B, 2 * C, generator=g) * 0.3).to(DEV)
    pad = (K * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV)
    lens_d = lens.to(DEV)
    y = torch.full((B, L, C), float("nan"), device=DEV)
    want_stats = tile != 64064
    st = ops.new_stats(B, L, C, DEV) if want_stats else None
    ops.conv_gemm(x.to(DEV), pc, y, dil=dil, pad=pad, lens_in=lens_d, lens_out=lens_d, pre=(sc.to(DEV), sh.to(DEV)), pre_act=ops.ACT_SNAKE,
                  pre_alpha=alpha.to(DEV), res=res.to(DEV), out_scale=0.5, tile=tile, stats=st)
    torch.cuda.synchronize()
    got = y.cpu()
    for b in range(B):
        n = int(lens[b])
        t = x[b:b + 1, :n].double() * sc[b].double() + sh[b].double()
        t = t + (1.0 / alpha.double()) * torch.sin(alpha.double() * t) ** 2
        ref = (ref_conv_nlc(t, w, bias, dil, pad)[0] + res[b, :n].double()) * 0.5
        assert rel_err(got[b, :n], ref) < 3e-5
        assert torch.isnan(got[b, n:]).all()  # rows beyond the item's length are never written
    if want_stats:
        ysafe = torch.nan_to_num(y)
        s1, h1 = ops.adain_from_partials(st, L, gb, lens_d)
        s2, h2 = ops.adain_coef(ysafe, gb, lens_d)
        torch.cuda.synchronize()
        assert rel_err(s1[:, :C], s2[:, :C]) < 2e-5 and rel_err(h1[:, :C], h2[:, :C]) < 2e-5


def _fq_ref(t, nmn, mx):
    """kitten_tts/quant.py:4-20 in float32, op by op (t float32 [L, C]; nmn = -min, mx = max, both joined with 0)."""
    mn = -np.float32(nmn)
    scale = (np.float32(mx) - mn) / np.float32(255.0)
    if scale == 0:
        return torch.zeros_like(t)
    zp = np.clip(np.rint(-mn / scale), 0, 255).astype(np.float32)
    tn = t.numpy().astype(np.float32)
    q = np.clip(np.rint(tn / scale + zp), 0, 255).astype(np.float32)
    return torch.from_numpy(((q - zp) * scale).astype(np.float32))


@pytest.mark.parametrize("tile", [0, 6128128, 6128064, 64128, 64064, 2064128])
@pytest.mark.parametrize("act", ["none", "leaky", "snake"])
def test_conv_gemm_quantising_prologue(ops, tile, act):
    """mi355_conv_gemm_args.pre_fq: extrema pass + the dynamic uint8 fake quantisation inside the conv prologue == quantising the prologue's output
    first and convolving that.  Without a prologue the quantised values are the reference's bit for bit (the conv must then agree with the conv of
    the materialised mi355_fake_quant_u8 tensor at conv accuracy); with AdaIN + LeakyReLU / Snake the prologue value carries ~1e-6 of its own, so a
    value within that distance of a rounding boundary may land in the neighbouring bin: bounded below by the bin width it can move."""
    g = torch.Generator().manual_seed(41)
    C = 64 if tile in (6128064, 64064) else 128
    B, L, K, dil = 3, 2100, 7, 3
    lens = torch.tensor([2100, 64, 1031], dtype=torch.int32)
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, L, C, generator=g) * 2.0 + 0.3
    sc = torch.rand(B, C, generator=g) + 0.5
    sh = torch.randn(B, C, generator=g) * 0.3
    alpha = torch.rand(C, generator=g) + 0.5
    pad = (K * dil - dil) // 2
    pc = ops.pack_conv(w, bias, DEV)
    lens_d, xd = lens.to(DEV), x.to(DEV)
    kw = {}
    if act != "none":
        kw.update(pre=(sc.to(DEV), sh.to(DEV)))
        kw.update(dict(pre_act=ops.ACT_SNAKE, pre_alpha=alpha.to(DEV)) if act == "snake" else dict(pre_act=ops.ACT_LEAKY, pre_slope=0.2))
    mm = ops.fake_quant_extrema(xd, lens=lens_d, **kw)
    y = torch.full((B, L, C), float("nan"), device=DEV)
    ops.conv_gemm(xd, pc, y, dil=dil, pad=pad, lens_in=lens_d, lens_out=lens_d, tile=tile, pre_fq=mm, **kw)
    torch.cuda.synchronize()
    got, mmc = y.cpu(), mm.cpu()
    for b in range(B):
        n = int(lens[b])
        t = x[b, :n].double()
        if act != "none":
            t = t * sc[b].double() + sh[b].double()
            t = t + (1.0 / alpha.double()) * torch.sin(alpha.double() * t) ** 2 if act == "snake" else F.leaky_relu(t, 0.2)
        # the extrema: those of the prologue output (joined with 0) at fp32 accuracy of the prologue
        assert abs(float(mmc[b, 0]) - max(0.0, float(-t.min()))) <= 2e-5 * float(t.abs().max())
        assert abs(float(mmc[b, 1]) - max(0.0, float(t.max()))) <= 2e-5 * float(t.abs().max())
        q = _fq_ref(t.float(), mmc[b, 0], mmc[b, 1])
        ref = ref_conv_nlc(q[None].double(), w, bias, dil, pad)[0]
        err = (got[b, :n].double() - ref).abs()
        peak = float(ref.abs().max())
        step = (float(mmc[b, 1]) + float(mmc[b, 0])) / 255.0
        if act == "none":
            assert float(err.max()) < 3e-5 * peak
        else:  # a few inputs one bin off: each moves an output by at most step * |w|; the bulk stays at conv accuracy
            assert float(err.max()) <= 3e-5 * peak + 3 * step * float(w.abs().max()) and float(err.median()) < 3e-5 * peak
        assert torch.isnan(got[b, n:]).all()
    if act == "none":  # the materialised path on the same values
        xq = ops.fake_quant_u8(xd, lens=lens_d)
        y2 = torch.zeros((B, L, C), device=DEV)
        ops.conv_gemm(xq, pc, y2, dil=dil, pad=pad, lens_in=lens_d, lens_out=lens_d, tile=tile)
        torch.cuda.synchronize()
        for b in range(B):
            n = int(lens[b])
            assert rel_err(got[b, :n], y2[b, :n].cpu()) < 1e-6


@pytest.mark.parametrize("cin,cout,k,s,L,row_off,tile", [(128, 16, 8, 4, 700, 0, 6128064), (128, 16, 8, 4, 700, 1, 0), (512, 256, 20, 10, 53, 0, 0), (256, 128, 12, 6, 130, 1, 0), (64, 32, 4, 2, 9, 0, 0),
                                                         (512, 256, 20, 10, 153, 0, 6128128), (256, 128, 12, 6, 330, 1, 6128128),
                                                         (512, 256, 20, 10, 153, 0, 86128128), (256, 128, 12, 6, 330, 1, 86128128),
                                                         (512, 256, 20, 10, 53, 0, 2064128), (256, 128, 12, 6, 130, 1, 42064128),
                                                         # long enough for interior tiles: the polyphase fast epilogue (conv_epilogue_interior_up)
                                                         (512, 256, 20, 10, 700, 0, 6128128), (256, 128, 12, 6, 1500, 1, 0), (128, 64, 8, 4, 900, 0, 6128128)])
def test_conv_transpose_polyphase(ops, cin, cout, k, s, L, row_off, tile):
    g = torch.Generator().manual_seed(k * s)
    p = (k - s) // 2
    w_t = bf16r(torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin / s))  # mx.conv_transpose1d layout
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(2, L, cin, generator=g)
    lout = (L - 1) * s - 2 * p + k
    res = torch.randn(2, lout + row_off, cout, generator=g)
    pc = ops.pack_conv_transpose(w_t, bias, s, DEV)
    y = torch.full((2, lout + row_off, cout), float("nan"), device=DEV)
    kp = k // s
    ops.conv_gemm(x.to(DEV), pc, y, pad=kp - 1, lout=L + kp - 1, pre_act=ops.ACT_LEAKY, pre_slope=0.1, res=res.to(DEV),
                  up=dict(s=s, p=p, cout=cout, row_off=row_off, lout=lout), tile=tile)
    torch.cuda.synchronize()
    xa = F.leaky_relu(x.double(), 0.1).transpose(1, 2)
    ref = F.conv_transpose1d(xa, w_t.permute(2, 0, 1).double(), bias.double(), stride=s, padding=p).transpose(1, 2)
    assert ref.shape[1] == lout
    ref = ref + res[:, row_off:].double()
    got = y[:, row_off:].cpu()
    assert rel_err(got, ref) < 3e-5
    if row_off:
        assert torch.isnan(y[:, :row_off]).all()  # the left pad row is the caller's


def test_conv_flat_strided_small_cin(ops):
    """noise_convs[0]: Conv1d(22 -> 256, k=12, stride=6, padding=3) on the contiguous [L, 22] feature map."""
    g = torch.Generator().manual_seed(3)
    B, L, cin, cout, k, s, p = 2, 601, 22, 256, 12, 6, 3
    w = bf16r(torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, L, cin, generator=g)
    lout = (L + 2 * p - k) // s + 1
    pc = ops.pack_conv(w.reshape(cout, 1, k * cin), bias, DEV)
    y = torch.full((B, lout, cout), float("nan"), device=DEV)
    lens = torch.tensor([L, 301], dtype=torch.int32)
    louts = ((lens + 2 * p - k) // s + 1).to(torch.int32)
    ops.conv_gemm(x.to(DEV), pc, y, lens_in=lens.to(DEV), lens_out=louts.to(DEV),
                  flat=dict(ldx=s * cin, x_off=-p * cin, channels=cin))
    torch.cuda.synchronize()
    for b in range(B):
        n = int(lens[b])
        ref = F.conv1d(x[b:b + 1, :n].double().transpose(1, 2), w.permute(0, 2, 1).double(), bias.double(), stride=s, padding=p)
        ref = ref.transpose(1, 2)[0]
        assert ref.shape[0] == int(louts[b])
        assert rel_err(y[b, : ref.shape[0]].cpu(), ref) < 3e-5
    # K=1, 22 channels, ld=22 (noise_convs[1])
    w1 = bf16r(torch.randn(128, 1, cin, generator=g))
    pc1 = ops.pack_conv(w1, None, DEV)
    y1 = torch.empty((B, L, 128), device=DEV)
    ops.conv_gemm(x.to(DEV), pc1, y1)
    torch.cuda.synchronize()
    assert rel_err(y1.cpu(), x.double() @ w1[:, 0].double().t()) < 3e-5


@pytest.mark.parametrize("tile", [0, 128128, 6128128, 86128128, 2064128, 32064128])
def test_conv_gemm_fused_instnorm_statistics(ops, tile):
    """Instance-norm statistics of the conv OUTPUT produced by the epilogue (stats=) + adain_from_partials must equal the
    separate pass (adain_coef) over the stored tensor: ragged batch, residual + scaling in the epilogue, large mean / std."""
    g = torch.Generator().manual_seed(21)
    B, L, C, K = 3, 700, 128, 3
    lens = torch.tensor([700, 65, 333], dtype=torch.int32).to(DEV)
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1 + 7.0  # mean >> std: the shifted single-pass variance must not cancel
    x = torch.randn(B, L, C, generator=g).to(DEV)
    res = torch.randn(B, L, C, generator=g).to(DEV)
    gb = (torch.randn(B, 2 * C, generator=g) * 0.3).to(DEV)
    pc = ops.pack_conv(w, bias, DEV)
    y = torch.zeros(B, L, C, device=DEV)
    st = ops.new_stats(B, L, C, DEV)
    st.fill_(float("nan"))
    ops.conv_gemm(x, pc, y, pad=1, lens_in=lens, lens_out=lens, res=res, out_scale=0.7, tile=tile, stats=st)
    sc, sh = ops.adain_from_partials(st, L, gb, lens)
    sc_ref, sh_ref = ops.adain_coef(y, gb, lens)
    torch.cuda.synchronize()
    assert torch.isfinite(sc).all() and torch.isfinite(sh).all()
    assert rel_err(sc[:, :C], sc_ref[:, :C]) < 2e-5, rel_err(sc[:, :C], sc_ref[:, :C])
    assert rel_err(sh[:, :C], sh_ref[:, :C]) < 2e-5, rel_err(sh[:, :C], sh_ref[:, :C])
    # and against float64 on the host
    for b in range(B):
        n = int(lens[b])
        v = y[b, :n].double().cpu()
        mean, var = v.mean(0), v.var(0, unbiased=False)
        want_sc = (1 + gb[b, :C].double().cpu()) / torch.sqrt(var + 1e-5)
        assert rel_err(sc[b, :C].cpu(), want_sc) < 2e-5
        assert rel_err(sh[b, :C].cpu(), gb[b, C:].double().cpu() - mean * want_sc) < 2e-5


@pytest.mark.parametrize("B,L,C", [(9, 2100, 128), (8, 40000, 200), (16, 70, 64)])
def test_adain_from_partials_batch_kernel(ops, B, L, C):
    """B >= 8 takes the bandwidth-shaped instantiation (16 channels per workgroup, the partials kept in registers between the mean and the M2 pass;
    L = 40 000 -> 625 blocks > 32 x 16 lanes: the re-reading fallback): same coefficients as float64 on the host, ragged lengths included."""
    g = torch.Generator().manual_seed(B * 1000 + C)
    lens = torch.randint(max(1, L // 3), L + 1, (B,), generator=g).to(torch.int32)
    lens[0] = L
    y = (torch.randn(B, L, C, generator=g) * 2.0 + 5.0)
    gb = torch.randn(B, 2 * C, generator=g) * 0.3
    nblk = (L + 63) // 64
    st = torch.full((B, nblk, C, 2), float("nan"))
    for b in range(B):
        n = int(lens[b])
        for e in range((n + 63) // 64):
            blk = y[b, e * 64:min(n, (e + 1) * 64)].double()
            st[b, e, :, 0] = blk.sum(0).float()
            st[b, e, :, 1] = ((blk - blk.mean(0)) ** 2).sum(0).float()
    sc, sh = ops.adain_from_partials(st.to(DEV), L, gb.to(DEV), lens.to(DEV))
    torch.cuda.synchronize()
    for b in range(B):
        v = y[b, : int(lens[b])].double()
        mean, var = v.mean(0), v.var(0, unbiased=False)
        want_sc = (1 + gb[b, :C].double()) / torch.sqrt(var + 1e-5)
        assert rel_err(sc[b, :C].cpu(), want_sc) < 2e-6
        assert rel_err(sh[b, :C].cpu(), gb[b, C:].double() - mean * want_sc) < 2e-6


def test_adain_coef(ops):
    g = torch.Generator().manual_seed(11)
    B, L, C = 3, 1000, 514
    lens = torch.tensor([1000, 333, 17], dtype=torch.int32)
    ld = ops.round_up(C, 32)
    x = torch.randn(B, L, ld, generator=g) * 3 + 50.0  # large mean: exercises the cancellation-safe path
    gb = torch.randn(B, 2 * C + 8, generator=g)
    sc, sh = ops.adain_coef(x.to(DEV)[:, :, :C], gb.to(DEV), lens=lens.to(DEV))
    torch.cuda.synchronize()
    for b in range(B):
        xb = x[b, : int(lens[b]), :C].double()
        mean, var = xb.mean(0), xb.var(0, unbiased=False)
        scale = (1 + gb[b, :C].double()) / torch.sqrt(var + 1e-5)
        shift = gb[b, C:2 * C].double() - mean * scale
        assert rel_err(sc[b, :C].cpu(), scale) < 2e-5
        assert float((sh[b, :C].cpu().double() - shift).abs().max() / shift.abs().max()) < 2e-5
    assert float(sc[:, C:].abs().max()) == 0.0


def test_layernorm_variants(ops):
    g = torch.Generator().manual_seed(5)
    B, L, C = 2, 37, 768
    x = torch.randn(B, L, C, generator=g) * 2 + 1
    r = torch.randn(B, L, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = torch.empty(B, L, C, device=DEV)
    ops.layernorm(x.to(DEV), y, weight=w.to(DEV), bias=b.to(DEV), res=r.to(DEV), eps=1e-12)
    torch.cuda.synchronize()
    ref = F.layer_norm((x + r).double(), (C,), w.double(), b.double(), 1e-12)
    assert float((y.cpu().double() - ref).abs().max()) < 2e-5
    gb = torch.randn(B, 2 * 512, generator=g)
    x2 = torch.randn(B, L, 640, generator=g)
    xd = x2.to(DEV)
    ops.layernorm(xd[:, :, :512], xd[:, :, :512], ada_gb=gb.to(DEV), eps=1e-5, post_act=ops.ACT_LEAKY, post_slope=0.2)
    torch.cuda.synchronize()
    ref = F.layer_norm(x2[:, :, :512].double(), (512,), None, None, 1e-5)
    ref = (1 + gb[:, None, :512].double()) * ref + gb[:, None, 512:].double()
    ref = torch.where(ref > 0, ref, ref * 0.2)
    assert float((xd[:, :, :512].cpu().double() - ref).abs().max()) < 2e-5
    assert torch.equal(xd[:, :, 512:].cpu(), x2[:, :, 512:])


@pytest.mark.parametrize("oct", ["1", "0"])
@pytest.mark.parametrize("H,In,L,B", [(256, 640, 41, 2), (128, 256, 23, 3), (64, 96, 19, 3)])
def test_lstm_vs_oracle(ops, monkeypatch, H, In, L, B, oct):
    """Both recurrence kernels of csrc/lstm.hip: lstm_oct_kernel (the default for H >= 64) and lstm_kernel (MI355_LSTM_OCT=0)."""
    from oracle import kokoro_ref

    monkeypatch.setenv("MI355_LSTM_OCT", oct)

    g = torch.Generator().manual_seed(H)
    s = 1 / math.sqrt(H)
    wts = {}
    for d in ("forward", "backward"):
        wts[f"l.Wx_{d}"] = bf16r((torch.rand(4 * H, In, generator=g) * 2 - 1) * s)
        wts[f"l.Wh_{d}"] = bf16r((torch.rand(4 * H, H, generator=g) * 2 - 1) * s)
        wts[f"l.bias_ih_{d}"] = bf16r((torch.rand(4 * H, generator=g) * 2 - 1) * s)
        wts[f"l.bias_hh_{d}"] = bf16r((torch.rand(4 * H, generator=g) * 2 - 1) * s)
    x = torch.randn(B, L, In, generator=g)
    lens = torch.tensor([L, max(1, L // 2), max(1, L - 3)][:B], dtype=torch.int32)
    wx = torch.cat([wts["l.Wx_forward"], wts["l.Wx_backward"]], 0)
    bias = torch.cat([wts["l.bias_ih_forward"] + wts["l.bias_hh_forward"], wts["l.bias_ih_backward"] + wts["l.bias_hh_backward"]])
    pc = ops.pack_conv(wx, bias, DEV)
    wh = ops.pack_lstm_wh(wts["l.Wh_forward"], wts["l.Wh_backward"], DEV)
    xp = torch.empty(B, L, 8 * H, device=DEV)
    lens_d = lens.to(DEV)
    ops.conv_gemm(x.to(DEV), pc, xp, lens_in=lens_d, lens_out=lens_d)
    out = torch.zeros(B, L, 2 * H + 32, device=DEV)
    ops.lstm_bidir(xp, wh, H, out[:, :, : 2 * H], lens=lens_d)
    torch.cuda.synchronize()
    p = kokoro_ref.P(wts, "l.", dtype=torch.float64)
    for b in range(B):
        n = int(lens[b])
        ref = kokoro_ref.bilstm(p, x[b:b + 1, :n].double())[0]
        assert float((out[b, :n, : 2 * H].cpu().double() - ref).abs().max()) < 5e-5
    assert float(out[:, :, 2 * H:].abs().max()) == 0.0
    # the same bf16 weights as an IEEE-half image scaled by a power of two (what the engines hand over: half the VALU work): every fp32 product and
    # sum is the bf16 image's times 2^k, so the output must be IDENTICAL, bit for bit, quantised hidden state or not
    scaled = ops.pack_lstm_wh_scaled(wts["l.Wh_forward"], wts["l.Wh_backward"], DEV)
    assert scaled is not None and scaled[1] > 0 and math.log2(scaled[1]) == round(math.log2(scaled[1]))
    for qh in (False, True):
        a = torch.zeros(B, L, 2 * H, device=DEV)
        bb = torch.zeros(B, L, 2 * H, device=DEV)
        ops.lstm_bidir(xp, wh, H, a, lens=lens_d, quant_h=qh)
        ops.lstm_bidir(xp, scaled[0], H, bb, lens=lens_d, quant_h=qh, wh_f16=True, wh_scale=scaled[1])
        torch.cuda.synchronize()
        assert torch.equal(a, bb), (qh, float((a - bb).abs().max()))
    # recurrent weights as IEEE half (precision 4 of the StyleTTS engines: float32 checkpoints): fp16-representable Wh, same bars
    wts16 = dict(wts)
    for d in ("forward", "backward"):
        wts16[f"l.Wh_{d}"] = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * s).half().float()
    wh16 = ops.pack_lstm_wh(wts16["l.Wh_forward"], wts16["l.Wh_backward"], DEV, f16=True)
    out16 = torch.zeros(B, L, 2 * H, device=DEV)
    ops.lstm_bidir(xp, wh16, H, out16, lens=lens_d, wh_f16=True)
    torch.cuda.synchronize()
    p16 = kokoro_ref.P(wts16, "l.", dtype=torch.float64, param_dtype=torch.float32)
    for b in range(B):
        n = int(lens[b])
        ref = kokoro_ref.bilstm(p16, x[b:b + 1, :n].double())[0]
        assert float((out16[b, :n].cpu().double() - ref).abs().max()) < 5e-5


def test_attention(ops):
    g = torch.Generator().manual_seed(2)
    B, T, heads, dh = 2, 83, 12, 64
    D = heads * dh
    qkv = torch.randn(B, T, 3 * D, generator=g)
    lens = torch.tensor([83, 40], dtype=torch.int32)
    out = torch.zeros(B, T, D, device=DEV)
    ops.attention(qkv.to(DEV), heads, dh, out, lens=lens.to(DEV))
    torch.cuda.synchronize()
    for b in range(B):
        n = int(lens[b])
        q, k, v = [t.view(n, heads, dh).transpose(0, 1).double() for t in qkv[b, :n].split(D, dim=-1)]
        ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).transpose(0, 1).reshape(n, D)
        assert float((out[b, :n].cpu().double() - ref).abs().max()) < 2e-5


def test_glue_kernels(ops):
    g = torch.Generator().manual_seed(9)
    B, T, C = 2, 20, 96
    table = torch.randn(50, C, generator=g)
    pos = torch.randn(T, C, generator=g)
    row = torch.randn(C, generator=g)
    idx = torch.randint(0, 50, (B, T), generator=g, dtype=torch.int32)
    lens = torch.tensor([20, 11], dtype=torch.int32)
    y = torch.full((B, T, C), float("nan"), device=DEV)
    ops.gather_rows(table.to(DEV), idx.to(DEV), y, pos_table=pos.to(DEV), add_row=row.to(DEV), lens=lens.to(DEV))
    torch.cuda.synchronize()
    ref = table[idx.long()] + pos[None] + row
    ref[1, 11:] = 0
    assert torch.equal(y.cpu(), ref)  # exact: same op order as words + position + token_type
    # duration head + alignment index (bit-exact integer path)
    logits = torch.randn(B, T, 64, generator=g) * 2
    dur, raw, frames, aidx = ops.duration_align(logits.to(DEV)[:, :, :50], T, B, 0.9, 2048, DEV, lens=lens.to(DEV))
    torch.cuda.synchronize()
    r = (torch.sigmoid(logits[:, :, :50]).sum(-1) / 0.9)
    margin = (r - torch.floor(r) - 0.5).abs().min()
    refd = torch.clamp(torch.round(r), 1, 100).to(torch.int32)
    refd[1, 11:] = 0
    assert float((raw.cpu() - r)[0].abs().max()) < 1e-4
    if margin > 1e-3:
        assert torch.equal(dur.cpu(), refd)
    d = dur.cpu()
    for b in range(B):
        want = torch.repeat_interleave(torch.arange(T), d[b].long())
        assert int(frames[b]) == want.numel()
        assert torch.equal(aidx[b, : want.numel()].cpu().long(), want)
    forced = torch.randint(1, 6, (B, T), generator=g, dtype=torch.int32)
    d2, _, fr2, idx2 = ops.duration_align(None, T, B, 1.0, 2048, DEV, forced=forced.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(d2.cpu(), forced) and int(fr2[0]) == int(forced[0].sum())
    # style broadcast
    v = torch.randn(B, 40, generator=g)
    yb = torch.zeros(B, T, 128, device=DEV)
    ops.broadcast_rows(v.to(DEV), yb[:, :, 88:128], lens=lens.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(yb[0, :, 88:].cpu(), v[0].expand(T, 40)) and float(yb[1, 11:].abs().max()) == 0
    # scalar strided conv
    x1 = torch.randn(B, 41, generator=g)
    yc = torch.zeros(B, 21, 8, device=DEV)
    ops.conv1d_c1_k3s2(x1.to(DEV), [0.3, -1.2, 0.7], 0.05, yc, 5)
    torch.cuda.synchronize()
    refc = F.conv1d(x1[:, None], torch.tensor([[[0.3, -1.2, 0.7]]]), torch.tensor([0.05]), stride=2, padding=1)[:, 0]
    assert float((yc[:, :, 5].cpu() - refc).abs().max()) < 1e-6


def test_adain_pool_up2(ops):
    from oracle import kokoro_ref

    g = torch.Generator().manual_seed(13)
    B, L, C = 2, 23, 96
    x = torch.randn(B, L, C, generator=g)
    sc, sh = torch.rand(B, C, generator=g) + 0.5, torch.randn(B, C, generator=g)
    w = torch.randn(C, 3, generator=g)
    bias = torch.randn(C, generator=g)
    y = torch.empty(B, 2 * L, C, device=DEV)
    ops.adain_pool_up2(x.to(DEV), sc.to(DEV), sh.to(DEV), 0.2, w.to(DEV), bias.to(DEV), y)
    torch.cuda.synchronize()
    a = F.leaky_relu(x.double() * sc[:, None].double() + sh[:, None].double(), 0.2).transpose(1, 2)
    ref = kokoro_ref.conv_transpose1d_mlx(a, w[:, :, None].double(), bias.double(), stride=2, padding=0, groups=C)[:, :, 1:]
    assert ref.shape[2] == 2 * L
    assert float((y.cpu().double() - ref.transpose(1, 2)).abs().max()) < 1e-5


@pytest.mark.parametrize("F2", [14, 3, 80])
def test_sine_source_and_stft_features(ops, F2):
    """SineGen path: contraction-free fp32 mirror of the reference's op order -> near bit-exact."""
    from oracle import kokoro_ref

    rng = np.random.default_rng(F2)
    B, up, H = 2, 300, 9
    f0 = (rng.uniform(-40, 400, size=(B, F2))).astype(np.float32)
    f0[0, : F2 // 3] = 0.0
    rand_ini = rng.uniform(size=(B, H)).astype(np.float32)
    noise = rng.standard_normal((B, F2 * up, H)).astype(np.float32)
    lw = rng.standard_normal((1, H)).astype(np.float32)
    lb = np.float32(0.03)
    w = {"m_source.l_linear.weight": torch.from_numpy(lw), "m_source.l_linear.bias": torch.tensor([lb])}
    ref = kokoro_ref.sine_source(kokoro_ref.P(w), torch.from_numpy(f0), rand_ini, noise, upsample=up)
    got = ops.sine_source(torch.from_numpy(f0).to(DEV), torch.from_numpy(rand_ini).to(DEV), torch.from_numpy(noise).to(DEV),
                          torch.from_numpy(lw[0]).to(DEV), float(lb), up)
    torch.cuda.synchronize()
    err = np.abs(got.cpu().numpy() - ref)
    assert err.max() < 2e-4, err.max()
    assert np.mean(err) < 2e-6
    # STFT features of the oracle's source (so both sides see identical input)
    win = torch.from_numpy(__import__("oracle.dsp_ref", fromlist=["x"]).hanning(20, periodic=True))
    y = torch.empty(B, F2 * up // 5 + 1, 22, device=DEV)
    ops.stft_magphase(torch.from_numpy(ref).to(DEV), 20, 5, win.to(DEV), y)
    torch.cuda.synchronize()
    mp = kokoro_ref.stft_mag_phase(ref, 20, 5).transpose(0, 2, 1)  # [B, frames, 22]
    g = y.cpu().numpy()
    assert np.abs(g[:, :, :11] - mp[:, :, :11]).max() < 1e-5
    dphi = np.abs(g[:, :, 11:] - mp[:, :, 11:])
    dphi = np.minimum(dphi, 2 * np.pi - dphi)  # +pi / -pi are the same angle
    big = mp[:, :, :11] > 1e-4
    assert dphi[big].max() < 1e-3


def test_istft_head(ops):
    from oracle import dsp_ref, kokoro_ref

    rng = np.random.default_rng(4)
    B, Fr = 2, 233
    x = rng.standard_normal((B, Fr, 22)).astype(np.float32)
    x[:, :, :11] = x[:, :, :11] * 0.5 - 1.0
    win = dsp_ref.hanning(20, periodic=True)
    audio = torch.full((B, (Fr - 1) * 5), float("nan"), device=DEV)
    xd = torch.zeros(B, Fr, 24, device=DEV)
    xd[:, :, :22] = torch.from_numpy(x).to(DEV)
    ops.istft_head(xd[:, :, :22], 20, 5, torch.from_numpy(win).to(DEV), audio)
    torch.cuda.synchronize()
    ref = kokoro_ref.istft_head(torch.from_numpy(x.transpose(0, 2, 1)), 20, 5)[:, 0].numpy()
    assert ref.shape == (B, (Fr - 1) * 5)
    assert np.abs(audio.cpu().numpy() - ref).max() < 2e-6


def _centered_frames(L, n_fft, hop):
    return 1 + (L + 2 * (n_fft // 2) - n_fft) // hop


@pytest.mark.parametrize("n_fft,hop,L", [(400, 160, 16000), (1024, 256, 12000), (20, 5, 3000), (96, 24, 1000), (56, 14, 700)])
def test_dsp_stft_istft_roundtrip_and_oracle(ops, n_fft, hop, L):
    from oracle import dsp_ref

    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal((2, L)).astype(np.float32)
    win = dsp_ref.hanning(n_fft)  # symmetric, like stft("hann")
    nfr = _centered_frames(L, n_fft, hop)
    spec = ops.stft_frames(torch.from_numpy(x).to(DEV), n_fft, hop, torch.from_numpy(win).to(DEV), 1, nfr)
    torch.cuda.synchronize()
    ref = np.stack([dsp_ref.stft(r, n_fft=n_fft, hop_length=hop, window=win) for r in x])
    scale = np.abs(ref).max()
    assert np.abs(spec.cpu().numpy() - ref).max() / scale < 2e-6
    # inverse (dsp.istft semantics: periodic window, w^2 normalisation) of the oracle spectrum
    wi = dsp_ref.hanning(n_fft + 1)[:-1]
    ola = (nfr - 1) * hop + n_fft
    norm = np.zeros(ola, np.float32)
    for f in range(nfr):
        norm[f * hop: f * hop + n_fft] += (wi * wi).astype(np.float32)
    out_len = ola - n_fft
    rec = ops.istft_frames(torch.from_numpy(ref).to(DEV), n_fft, hop, torch.from_numpy(wi).to(DEV),
                           torch.from_numpy(norm).to(DEV), 1, False, n_fft // 2, out_len)
    torch.cuda.synchronize()
    want = np.stack([dsp_ref.istft(r.T, hop_length=hop, win_length=n_fft, window=wi, normalized=True) for r in ref])
    assert want.shape[1] == out_len
    assert np.abs(rec.cpu().numpy() - want).max() < 5e-5


def test_logmel_golden_through_gpu(ops, golden):
    """The reference's own STFT+mel golden vectors (test_qwen3_tts.py:175-353) through the HIP path."""
    from oracle import dsp_ref

    g = golden["qwen3_mel_spectrogram"]
    np.random.seed(42)
    audio = np.random.randn(12000).astype(np.float32)
    pad = (1024 - 256) // 2
    padded = np.concatenate([audio[1: pad + 1][::-1], audio, audio[-(pad + 1): -1][::-1]])
    fb = dsp_ref.mel_filters(24000, 1024, 128, 0.0, 12000.0, norm="slaney", mel_scale="slaney")
    win = dsp_ref.hanning(1024)
    nfr = 1 + (len(padded) - 1024) // 256
    mel = ops.logmel(torch.from_numpy(padded[None].copy()).to(DEV), 1024, 256, torch.from_numpy(win).to(DEV), 0, nfr,
                     torch.from_numpy(fb).to(DEV), 1)
    torch.cuda.synchronize()
    m = mel.cpu().numpy()
    assert list(m.shape) == g["shape"]
    kw = dict(rtol=g["rtol"], atol=g["atol"])
    np.testing.assert_allclose(m[0, 0, g["bins"]], g["frame0"], **kw)
    np.testing.assert_allclose(m[0, 23, g["bins"]], g["frame23"], **kw)
    np.testing.assert_allclose(m[0, -1, g["bins"]], g["frame_last"], **kw)
    np.testing.assert_allclose(m.mean(), g["mean"], **kw)
    np.testing.assert_allclose(m.std(), g["std"], **kw)
    assert np.abs(m - dsp_ref.qwen3_mel_spectrogram(audio)).max() < 2e-4
    # whisper front end: 3 s of noise + 1 s of zero padding
    a = np.random.default_rng(0).standard_normal(48000).astype(np.float32)
    ap = np.concatenate([a, np.zeros(16000, np.float32)])
    fbw = dsp_ref.mel_filters(16000, 400, 80, norm="slaney", mel_scale=None)
    nf = _centered_frames(len(ap), 400, 160) - 1  # whisper drops the last frame
    w = ops.logmel(torch.from_numpy(ap[None].copy()).to(DEV), 400, 160, torch.from_numpy(dsp_ref.hanning(400)).to(DEV), 1, nf,
                   torch.from_numpy(fbw).to(DEV), 0)
    torch.cuda.synchronize()
    ref = dsp_ref.whisper_log_mel(a, padding=16000)
    assert ref.shape == tuple(w.shape[1:])
    assert np.abs(w.cpu().numpy()[0] - ref).max() < 2e-4


@pytest.mark.parametrize("kw", [dict(),
                                dict(sample_rate=16000, win_len=400, win_inc=160, num_mels=80, win_type="povey", snip_edges=False),
                                dict(sample_rate=16000, win_len=400, win_inc=160, num_mels=80, win_type="hanning", preemphasis=0.0),
                                dict(sample_rate=16000, win_len=512, win_inc=128, num_mels=23, win_type="rectangular", low_freq=0.0, high_freq=-200.0,
                                     snip_edges=False)])
def test_compute_fbank_kaldi_matches_oracle(ops, kw):
    """dsp.compute_fbank_kaldi (dsp.py:898-997) on the GPU: framing kernel + fused FFT/power/mel/log vs the numpy restatement; the
    reference's own pins are the shapes ([58, 60] default case, sts/tests/test_mossformer2_se.py:134-208)."""
    from mlx_audio_amd import dsp
    from oracle import dsp_ref

    np.random.seed(42)
    audio = (np.random.randn(24000) * 3000.0).astype(np.float32)   # int16-range amplitudes as the callers pass them (fireredasr2.py:544)
    if not kw.get("snip_edges", True):   # whole number of shifts: otherwise the reference's strided view leaves its padded buffer (undefined)
        with pytest.raises(Exception, match="reflected edges"):
            dsp.compute_fbank_kaldi(torch.from_numpy(audio[:23999 - 23999 % kw["win_inc"] + kw["win_inc"] - 1]), dither=0.0, **kw)
        audio = audio[:24000 - 24000 % kw["win_inc"]]
    want = dsp_ref.compute_fbank_kaldi(audio, dither=0.0, **kw)
    got = dsp.compute_fbank_kaldi(torch.from_numpy(audio), dither=0.0, **kw)
    torch.cuda.synchronize()
    assert tuple(got.shape) == want.shape
    if not kw:
        assert tuple(got.shape) == (58, 60)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-4
    # explicit dither noise: same draws on both sides
    noise = np.random.default_rng(3).standard_normal((want.shape[0], int(kw.get("win_len", 1920)))).astype(np.float32)
    want_d = dsp_ref.compute_fbank_kaldi(audio, dither=1.0, noise=noise, **kw)
    got_d = dsp.compute_fbank_kaldi(torch.from_numpy(audio), dither=1.0, noise=torch.from_numpy(noise), **kw)
    assert np.abs(got_d.cpu().numpy() - want_d).max() < 2e-4
    # default dither draws on the device: finite, same shape, close to the undithered features at this amplitude
    d = dsp.compute_fbank_kaldi(torch.from_numpy(audio), **kw)
    assert tuple(d.shape) == want.shape and torch.isfinite(d).all()
    # too-short input under snip_edges
    assert tuple(dsp.compute_fbank_kaldi(torch.from_numpy(audio[:100])).shape) == (0, 60)
    assert tuple(dsp.compute_fbank_kaldi(torch.from_numpy(audio[None])[:, :4800], dither=0.0).shape) == (8, 60)   # [1, L] input accepted


@pytest.mark.parametrize("n_fft,hop,n_mels,mode,pad_mode,L,B", [
    (400, 160, 80, 0, 1, 48000, 3),      # Whisper geometry; 301 frames = 12 full tiles of 24 + a ragged one
    (400, 160, 80, 0, 2, 5000, 2),       # constant padding, short signal (every tile touches an edge)
    (1024, 256, 128, 1, 0, 12768, 2),    # Qwen3 speaker mel (caller pads): sqrt(|X|^2 + 1e-9), natural log
    (1024, 256, 100, 3, 1, 24000, 1),    # Vocos mel
    (512, 512, 23, 2, 0, 512 * 37, 1),   # Kaldi fbank frames (hop = n_fft, no overlap)
    (512, 128, 60, 2, 1, 9000, 2),
])
def test_fast_stft_logmel_equals_lds_stockham_and_oracle(ops, monkeypatch, n_fft, hop, n_mels, mode, pad_mode, L, B):
    """The register-resident two-pass kernels (csrc/fft_fast.h: n_fft 400 / 512 / 1024) against (i) the numpy restatement of dsp.stft
    (dsp.py:385-433) and (ii) the LDS Stockham kernel they replace on these sizes (MI355_FFT_FAST=0), for the complex spectrum and for every
    mel mode; then a DENSE filterbank (no zero spans: the rows do not fit the LDS budget and are read from L2 instead)."""
    from oracle import dsp_ref

    rng = np.random.default_rng(n_fft + hop + L)
    x = (rng.standard_normal((B, L)) * np.linspace(0.2, 3.0, L)[None]).astype(np.float32)
    win = dsp_ref.hanning(n_fft)
    nfr = (1 + (L + (2 * (n_fft // 2) if pad_mode else 0) - n_fft) // hop)
    xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(win).to(DEV)
    sr = {400: 16000, 512: 16000, 1024: 24000}[n_fft]
    fb = dsp_ref.mel_filters(sr, n_fft, n_mels, norm="slaney", mel_scale="slaney")
    fbd = torch.from_numpy(np.ascontiguousarray(fb)).to(DEV)
    dense = torch.from_numpy((rng.random((n_mels, n_fft // 2 + 1)) * 1e-2 + 1e-4).astype(np.float32)).to(DEV)

    def run():
        spec = ops.stft_frames(xd, n_fft, hop, wd, pad_mode, nfr)
        mel = ops.logmel(xd, n_fft, hop, wd, pad_mode, nfr, fbd, mode)
        mel_dense = ops.logmel(xd, n_fft, hop, wd, pad_mode, nfr, dense, mode)
        torch.cuda.synchronize()
        return spec.cpu().numpy(), mel.cpu().numpy(), mel_dense.cpu().numpy()

    monkeypatch.setenv("MI355_FFT_FAST", "1")
    s1, m1, d1 = run()
    monkeypatch.setenv("MI355_FFT_FAST", "0")
    s0, m0, d0 = run()
    scale = np.abs(s0).max()
    assert np.abs(s1 - s0).max() / scale < 2e-6
    if pad_mode == 1:
        ref = np.stack([dsp_ref.stft(r, n_fft=n_fft, hop_length=hop, window=win) for r in x])
        assert ref.shape == s1.shape and np.abs(s1 - ref).max() / scale < 2e-6
    # log of a sum of powers: compare where the value is above the clamp floor by a margin, at the fp32 level of the power itself elsewhere
    assert np.isfinite(m1).all() and np.abs(m1 - m0).max() < 2e-4, np.abs(m1 - m0).max()
    assert np.abs(d1 - d0).max() < 2e-4, np.abs(d1 - d0).max()


@pytest.mark.parametrize("L,C,K,prec", [(256, 512, 3, 2), (264, 512, 3, 2), (5280, 256, 7, 5)])
def test_conv_ws4_fused_statistics_identical_rows_repeatable(ops, L, C, K, prec):
    """Regression (round 5): B IDENTICAL rows through the wave-specialised kernel (>= 128 tiles) with fused statistics, several launches.  Every row
    must carry the same partials, run after run, and they must be the statistics of the stored output.  With hipcc's SLP vectoriser on, the interior
    epilogue's packed-fp32 code (v_pk_fma_f32 with SGPR-pair operands) returned RANDOM M2 terms for a few 32-column fragments per launch; the
    library is built with -fno-slp-vectorize (mlx_audio_amd/build.py)."""
    g = torch.Generator().manual_seed(3)
    B = 64 if L < 1000 else 4
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    bias = torch.randn(C, generator=g) * 0.1
    pc = ops.pack_conv(w, bias, DEV, mx=prec == 5)
    x = torch.randn(1, L, C, generator=g).expand(B, -1, -1).contiguous().to(DEV)
    sc = (torch.rand(1, C, generator=g) + 0.5).expand(B, -1).contiguous().to(DEV)
    sh = (torch.randn(1, C, generator=g) * 0.1).expand(B, -1).contiguous().to(DEV)
    lens = torch.full((B,), L, dtype=torch.int32, device=DEV)
    nblk = (L + 63) // 64
    first = None
    for rep in range(4):
        y = torch.zeros(B, L, C, device=DEV)
        st = ops.new_stats(B, L, C, DEV)
        st.fill_(float("nan"))
        ops.conv_gemm(x, pc, y, pad=(K - 1) // 2, lens_in=lens, lens_out=lens, pre=(sc, sh), pre_act=ops.ACT_LEAKY, pre_slope=0.2, stats=st, precision=prec)
        torch.cuda.synchronize()
        assert torch.equal(y, y[0:1].expand_as(y)) and torch.equal(st, st[0:1].expand_as(st)), rep
        if first is None:
            first = (y.clone(), st.clone())
            ref = torch.empty_like(st[0])
            for e in range(nblk):
                blk = y[0, e * 64:min(L, (e + 1) * 64)].double()
                ref[e, :, 0] = blk.sum(0).float()
                ref[e, :, 1] = ((blk - blk.mean(0, keepdim=True)) ** 2).sum(0).float()
            assert float((st[0] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        else:
            assert torch.equal(y, first[0]) and torch.equal(st, first[1]), rep


@pytest.mark.parametrize("L,C,K", [(264, 512, 3), (5280, 128, 7)])
def test_conv_ws4_ext_partials_identical_rows_repeatable(ops, L, C, K):
    """The same gate for the per-block EXTREMA a quantising conv leaves (ext_partial, round 5): they are produced next to the fused statistics by the same
    xor-32 exchange pattern.  B identical rows, several launches: every row carries the same (min, max) partials, run after run, and they are the
    extrema of the stored output."""
    g = torch.Generator().manual_seed(4)
    B = 64 if L < 1000 else 4
    w = bf16r(torch.randn(C, K, C, generator=g) / math.sqrt(K * C))
    pc = ops.pack_conv(w, torch.randn(C, generator=g) * 0.1, DEV)
    x = torch.randn(1, L, C, generator=g).expand(B, -1, -1).contiguous().to(DEV)
    sc = (torch.rand(1, C, generator=g) + 0.5).expand(B, -1).contiguous().to(DEV)
    sh = (torch.randn(1, C, generator=g) * 0.1).expand(B, -1).contiguous().to(DEV)
    lens = torch.full((B,), L, dtype=torch.int32, device=DEV)
    kw = dict(pre=(sc, sh), pre_act=ops.ACT_LEAKY, pre_slope=0.2)
    assert ops.conv_ext_supported(x, pc, B=B, lout=L, pre_act=ops.ACT_LEAKY)
    mm = ops.fake_quant_extrema(x, lens=lens, **kw)
    nblk = (L + 63) // 64
    first = None
    for rep in range(4):
        y = torch.zeros(B, L, C, device=DEV)
        ext = ops.new_ext(B, L, C, DEV)
        ext.fill_(float("nan"))
        ops.conv_gemm(x, pc, y, pad=(K - 1) // 2, lens_in=lens, lens_out=lens, pre_fq=mm, ext=ext, **kw)
        torch.cuda.synchronize()
        assert torch.equal(y, y[0:1].expand_as(y)) and torch.equal(ext, ext[0:1].expand_as(ext)), rep
        if first is None:
            first = (y.clone(), ext.clone())
            for e in range(nblk):
                blk = y[0, e * 64:min(L, (e + 1) * 64)]
                assert torch.equal(ext[0, e, :, 0], blk.min(0).values) and torch.equal(ext[0, e, :, 1], blk.max(0).values), e
        else:
            assert torch.equal(y, first[0]) and torch.equal(ext, first[1]), rep


@pytest.mark.parametrize("mode,n_fft,hop,n_mels", [(0, 400, 160, 80), (4, 512, 160, 80), (2, 512, 512, 23), (3, 1024, 256, 100), (1, 1024, 256, 128)])
def test_fast_logmel_silent_tiles_are_bit_identical_to_the_full_path(ops, monkeypatch, mode, n_fft, hop, n_mels):
    """Tiles whose samples are all zero skip their transforms in csrc/fft_fast.h and write the clamp floor's logarithm directly.  The outputs must be
    EXACTLY what the full path writes for the same frames: checked by moving the silence (a signal with a silent middle and tail vs the same signal
    with 1e-30 added everywhere, which defeats the shortcut and changes no power above rounding) and against the LDS Stockham kernel."""
    from oracle import dsp_ref

    rng = np.random.default_rng(mode)
    L = 40 * n_fft
    x = np.zeros((2, L), np.float32)
    x[0, : 6 * n_fft] = rng.standard_normal(6 * n_fft)
    x[0, 17 * n_fft: 19 * n_fft + 7] = rng.standard_normal(2 * n_fft + 7) * 0.1
    x[1, 30 * n_fft:] = rng.standard_normal(10 * n_fft)
    win = dsp_ref.hanning(n_fft)
    sr = {400: 16000, 512: 16000, 1024: 24000}[n_fft]
    fb = torch.from_numpy(np.ascontiguousarray(dsp_ref.mel_filters(sr, n_fft, n_mels, norm="slaney", mel_scale="slaney"))).to(DEV)
    nfr = 1 + L // hop
    wd = torch.from_numpy(win).to(DEV)

    def run(sig):
        y = ops.logmel(torch.from_numpy(sig).to(DEV), n_fft, hop, wd, 1, nfr, fb, mode, log_guard=2.0 ** -24 if mode == 4 else 0.0)
        spec = ops.stft_frames(torch.from_numpy(sig).to(DEV), n_fft, hop, wd, 1, nfr)
        torch.cuda.synchronize()
        return y.cpu().numpy(), spec.cpu().numpy()

    monkeypatch.setenv("MI355_FFT_FAST", "1")
    y1, s1 = run(x)
    y2, s2 = run(x + np.float32(1e-30))           # no tile is all-zero any more; powers change by < 1e-50
    monkeypatch.setenv("MI355_FFT_FAST", "0")
    y0, s0 = run(x)
    silent = np.abs(s0).max(axis=2) == 0.0         # frames whose spectrum is exactly zero
    assert silent.sum() > nfr                      # the test signal really has silent frames
    assert np.array_equal(y1[silent], y2[silent]), "silent-tile shortcut differs from the full path on silent frames"
    assert np.abs(y1 - y2).max() < 1e-5 and np.abs(y1 - y0).max() < 2e-4
    # spectrum: a silent frame that shares its complex transform with a sounding one (two real frames ride one transform) carries the Hermitian split's
    # rounding residue of its partner (~1e-7 of the partner's magnitude) in the two-pass kernel; frames of all-silent TILES are exact zeros
    assert np.abs(s1 - s0).max() / np.abs(s0).max() < 2e-6
    assert (np.abs(s1).max(axis=2) == 0.0).sum() > nfr // 2
