"""Pre-split activations (include/mi355audio.h: x_split / y_split, mi355_split16, mi355_layernorm.y_split) through the C ABI:
 (a) the producers of SPLIT words -- the elementwise kernel, LayerNorm, a conv epilogue -- against the numpy statement of the format (oracle/mx_ref.py),
     bit for bit;
 (b) a launch that READS split words against the same launch on the float32 tensor: bit for bit (the words are exactly the two numbers the float32
     path's prologue makes of each value) -- GEMM mode and conv mode, ragged items, channel tails, both hi + lo precisions;
 (c) what the ABI refuses."""
import numpy as np
import pytest
import torch

from oracle import mx_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
WS = 6128128


@pytest.fixture(scope="module")
def ops():
    from mlx_audio_amd import ops as _ops

    _ops.require_gpu()
    return _ops


def words(t):
    return t.detach().contiguous().view(torch.int32).cpu().numpy().view(np.uint32)


def wide(shape, g):
    """values over ~12 binades with a few exact zeros, half-way cases and out-of-range magnitudes"""
    x = torch.randn(shape, generator=g) * torch.exp2(torch.randint(-9, 4, shape, generator=g).float())
    f = x.flatten()
    f[::97] = 0.0
    f[1::101] = 1.0 + 2.0 ** -11
    f[2::103] = 70000.0
    f[3::107] = -1.0e6
    return x


@pytest.mark.parametrize("fmt", [2, 4])
def test_split16_kernel_matches_the_format(ops, fmt):
    g = torch.Generator().manual_seed(fmt)
    x = wide((3, 130, 64), g)
    got = words(ops.split16(x.to(DEV), fmt))
    exp = mx_ref.split16_words(x.numpy(), fmt)
    assert np.array_equal(got, exp)
    # in place
    xd = x.to(DEV)
    ops.split16(xd, fmt, xd)
    assert np.array_equal(words(xd), exp)


@pytest.mark.parametrize("fmt", [2, 4])
def test_layernorm_split_output(ops, fmt):
    g = torch.Generator().manual_seed(10 + fmt)
    x = torch.randn((2, 77, 768), generator=g).to(DEV)
    w, b = (torch.rand(768, generator=g) + 0.5).to(DEV), (torch.randn(768, generator=g) * 0.2).to(DEV)
    y = ops.layernorm(x, torch.empty_like(x), weight=w, bias=b, eps=1e-5)
    ys = ops.layernorm(x, torch.empty_like(x), weight=w, bias=b, eps=1e-5, split=fmt)
    assert np.array_equal(words(ys), mx_ref.split16_words(y.cpu().numpy(), fmt))


def pack(ops, w, b, prec):
    return ops.pack_conv(w, b, DEV, f16=prec == 4)


@pytest.mark.parametrize("prec", [2, 4])
@pytest.mark.parametrize("cin,cout,k,dil,L,B,ragged", [
    (768, 384, 1, 1, 300, 3, True),     # GEMM mode: 64-channel super-chunks, ragged items, a 92-row last tile
    (800, 130, 1, 1, 257, 2, False),    # GEMM mode: channel tail inside the last super-chunk (800 = 12 x 64 + 32), column tail
    (96, 256, 1, 1, 200, 2, True),      # GEMM mode with an odd number of 32-channel chunks
    (128, 128, 3, 2, 400, 2, True),     # conv mode, no prologue: halo rows left and right, dilation
    (64, 64, 5, 1, 300, 2, False),      # conv mode on the 128 x 64 tile
])
def test_x_split_launch_is_bit_identical_to_the_float_launch(ops, prec, cin, cout, k, dil, L, B, ragged):
    g = torch.Generator().manual_seed(1000 * prec + cin + k)
    w = (torch.randn(cout, k, cin, generator=g) / (k * cin) ** 0.5).to(torch.bfloat16).float()
    bias = torch.randn(cout, generator=g) * 0.1
    pc = pack(ops, w, bias, prec)
    ld = ops.round_up(cin, 32)
    x = torch.zeros((B, L, ld))
    x[:, :, :cin] = wide((B, L, cin), g) * 0.25
    x = x.to(DEV)
    lens = torch.tensor([L, L - 37, 5][:B], dtype=torch.int32, device=DEV) if ragged else None
    res = torch.randn((B, L, cout), generator=g).to(DEV)
    pad = (k - 1) * dil // 2
    y0 = torch.full((B, L, cout), 7.0, device=DEV)
    y1 = torch.full((B, L, cout), 7.0, device=DEV)
    kw = dict(dil=dil, pad=pad, lens_in=lens, lens_out=lens, res=res, precision=prec)
    tile = WS if cout > 64 else 6128064   # (the float launch on the same kernel as the split one: the accumulation order is the kernel's)
    ops.conv_gemm(x[:, :, :cin], pc, y0, tile=tile, **kw)
    xs = ops.split16(x, prec)
    ops.conv_gemm(xs[:, :, :cin], pc, y1, x_split=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    # and the float launch is right: against float64 on the valid rows of item 0
    xr = x[0, :, :cin].double().cpu()
    xr = torch.clamp(xr, -65504, 65504) if prec == 4 else xr
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(xr.t()[None], (pad, (k - 1) * dil - pad)), w.permute(0, 2, 1).double(), bias.double(), dilation=dil)[0].t()
    ref = ref + res[0].double().cpu()
    err = float((y1[0].double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-5, err


@pytest.mark.parametrize("prec", [2, 4])
def test_y_split_epilogue_and_chain(ops, prec):
    """mlp1 -> GELU -> (split words) -> mlp2 + residual: the chained pair on split activations equals the float pair bit for bit, and the words the first
    launch stores are the format's words of the floats it would have stored."""
    g = torch.Generator().manual_seed(77 + prec)
    B, L, C, Hd = 2, 333, 256, 1024
    w1 = (torch.randn(Hd, 1, C, generator=g) / C ** 0.5).to(torch.bfloat16).float()
    w2 = (torch.randn(C, 1, Hd, generator=g) / Hd ** 0.5).to(torch.bfloat16).float()
    p1, p2 = pack(ops, w1, torch.randn(Hd, generator=g) * 0.1, prec), pack(ops, w2, torch.randn(C, generator=g) * 0.1, prec)
    x = torch.randn((B, L, C), generator=g).to(DEV)
    mid_f, mid_s = torch.empty((B, L, Hd), device=DEV), torch.empty((B, L, Hd), device=DEV)
    out_f, out_s = x.clone(), x.clone()
    ops.conv_gemm(x, p1, mid_f, post_act=ops.ACT_GELU, precision=prec, tile=WS)
    ops.conv_gemm(mid_f, p2, out_f, res=out_f, precision=prec, tile=WS)
    xs = ops.split16(x, prec)
    ops.conv_gemm(xs, p1, mid_s, post_act=ops.ACT_GELU, precision=prec, x_split=True, y_split=True)
    ops.conv_gemm(mid_s, p2, out_s, res=out_s, precision=prec, x_split=True)
    torch.cuda.synchronize()
    assert np.array_equal(words(mid_s), mx_ref.split16_words(mid_f.cpu().numpy(), prec))
    assert torch.equal(out_f, out_s)


def test_split_arguments_the_abi_refuses(ops):
    from mlx_audio_amd import _lib

    g = torch.Generator().manual_seed(5)
    w = torch.randn(128, 1, 128, generator=g).to(torch.bfloat16).float()
    pc2, pc3 = ops.pack_conv(w, None, DEV), ops.pack_conv(w, None, DEV, f16=True)
    x = torch.randn((1, 256, 128), generator=g).to(DEV)
    y = torch.empty((1, 256, 128), device=DEV)
    with pytest.raises(_lib.Mi355Error):   # a single-pass precision has no lo part to carry
        ops.conv_gemm(x, pc3, y, precision=3, x_split=True)
    sc = torch.ones((1, 128), device=DEV)
    with pytest.raises(_lib.Mi355Error):   # no prologue on split words
        ops.conv_gemm(x, pc2, y, pre=(sc, sc), precision=2, x_split=True)
    with pytest.raises(_lib.Mi355Error):   # y holds words: nothing to accumulate into
        ops.conv_gemm(x, pc2, y, precision=2, y_split=True, accumulate=True)
    with pytest.raises(_lib.Mi355Error):
        ops.split16(x, 3)
