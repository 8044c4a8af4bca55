"""Host side of conv precision 5: the MX weight image (csrc/api.cpp ``mi355_pack_conv_weight_mx_host``) against the independent numpy statement of
its layout (oracle/mx_ref.py), and the e4m3 helpers of that oracle against torch's float8_e4m3fn conversion."""
import math

import numpy as np
import pytest
import torch

from mlx_audio_amd import _lib
from oracle import mx_ref


def test_e4m3_round_matches_torch_float8():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(20000) * 60.0, rng.standard_normal(20000) * 0.01, np.array([0.0, 448.0, -448.0, 460.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10])])
    v = v[np.abs(v) <= 464.0]  # torch maps values that round beyond 448 to NaN; the oracle (and the device conversion) saturate: compared below the edge
    want = torch.from_numpy(v.astype(np.float32)).to(torch.float8_e4m3fn)
    got = mx_ref.e4m3_round(v.astype(np.float32))
    ok = ~torch.isnan(want.to(torch.float32)).numpy()
    assert np.array_equal(got[ok], want.to(torch.float32).numpy().astype(np.float64)[ok])
    assert np.array_equal(mx_ref.e4m3_bits(got[ok]), want.view(torch.uint8).numpy()[ok])


@pytest.mark.parametrize("cout,k,cin", [(128, 3, 128), (200, 7, 96), (64, 11, 40), (130, 1, 64), (256, 2, 33)])
def test_mx_image_matches_layout_statement(cout, k, cin):
    lib = _lib.load()
    g = torch.Generator().manual_seed(cout + k + cin)
    w = (torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin)).to(torch.bfloat16).to(torch.float32)
    w[min(5, cout - 1)] = 0.0   # an all-zero column: scale byte 127
    wn = w.numpy()
    nb = lib.mi355_packed_conv_weight_mx_bytes(cout, k, cin)
    out = np.full(nb, 0xAA, dtype=np.uint8)
    assert lib.mi355_pack_conv_weight_mx_host(wn.ctypes.data, cout, k, cin, out.ctypes.data) == 0
    want = mx_ref.pack_mx_image(wn)
    assert want.shape == out.shape
    assert np.array_equal(out, want)
    # the image's fp16 part holds the bf16-valued weights exactly, the e4m3 part within 2^-4 of each value relative to its own magnitude class
    wq = mx_ref.quantise_weights(wn)
    nz = np.abs(wn) > np.abs(wn).reshape(cout, -1).max(axis=1)[:, None, None] * 2.0 ** -6
    assert np.all(np.abs(wq - wn)[nz] <= np.abs(wn)[nz] * 2.0 ** -4)


def test_mx_pack_rejects_non_finite():
    lib = _lib.load()
    w = np.zeros((32, 3, 32), dtype=np.float32)
    w[3, 1, 2] = np.inf
    out = np.zeros(lib.mi355_packed_conv_weight_mx_bytes(32, 3, 32), dtype=np.uint8)
    assert lib.mi355_pack_conv_weight_mx_host(w.ctypes.data, 32, 3, 32, out.ctypes.data) != 0
    assert b"non-finite" in lib.mi355_last_error()


# ------------------------------------------------------------------------------------------------ precision 6: the MX4 image (FP4 lo slices)
def test_e2m1_grid_rounding_and_saturation():
    """The oracle's e2m1 rounding on the cases probed on the device (profiles/r6_mfma_fp4_probe_call2.jsonl): ties go to the EVEN code, |v| >= 6 saturates."""
    v = np.array([0.0, 0.24, 0.25, 0.26, 0.5, 0.74, 0.75, 0.76, 1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 7.0, 8.0, 100.0, -0.25, -0.75, -1.25, -2.5, -5.0, -7.0])
    want = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 4, 4, 4, 5, 6, 6, 6, 7, 7, 7, 7, 8, 10, 10, 12, 14, 15])   # low nibbles of the device's sel0 column
    assert np.array_equal(mx_ref.e2m1_code(v), want)
    assert np.array_equal(mx_ref.e2m1_round(np.array([0.26, -2.5, 5.0, 9.0])), np.array([0.5, -2.0, 4.0, 6.0]))


@pytest.mark.parametrize("cout,k,cin", [(128, 3, 128), (200, 7, 96), (64, 11, 40), (130, 1, 64), (256, 2, 33)])
def test_mx4_image_matches_layout_statement(cout, k, cin):
    lib = _lib.load()
    g = torch.Generator().manual_seed(cout + k + cin)
    w = (torch.randn(cout, k, cin, generator=g) / math.sqrt(k * cin)).to(torch.bfloat16).to(torch.float32)
    w[min(5, cout - 1)] = 0.0   # an all-zero column: scale byte 127
    wn = w.numpy()
    nb = lib.mi355_packed_conv_weight_mx_bytes(cout, k, cin)
    out = np.full(nb, 0xAA, dtype=np.uint8)
    assert lib.mi355_pack_conv_weight_mx4_host(wn.ctypes.data, cout, k, cin, out.ctypes.data) == 0
    want = mx_ref.pack_mx4_image(wn)
    assert want.shape == out.shape
    assert np.array_equal(out, want)
    # the fp16 part of the two images is the same; the 4-bit grid holds every weight within half a grid step of its column's scale
    out8 = np.zeros(nb, dtype=np.uint8)
    assert lib.mi355_pack_conv_weight_mx_host(wn.ctypes.data, cout, k, cin, out8.ctypes.data) == 0
    chunks, ntp, npair = (cin + 31) // 32, (cout + 127) // 128 * 4, (k + 1) // 2
    for ch in range(chunks):
        b = ch * (k + npair) * ntp * 2048
        assert np.array_equal(out[b:b + k * ntp * 2048], out8[b:b + k * ntp * 2048])
    wq = mx_ref.quantise_weights4(wn)
    sc = np.exp2(mx_ref.column_scale_exponents4(wn).astype(np.float64))[:, None, None]
    scb = np.broadcast_to(sc, wn.shape)
    err = np.abs(wq - wn)
    assert np.all(err < 2.0 * scb + 1e-30)                       # saturation: a scaled value in (6, 8) becomes 6
    assert np.all(err[np.abs(wn) <= 6 * scb] <= 1.0 * scb[np.abs(wn) <= 6 * scb] + 1e-30)    # inside the grid: half the largest step (4 .. 6)
    assert np.all(err[np.abs(wn) <= 2 * scb] <= 0.25 * scb[np.abs(wn) <= 2 * scb] + 1e-30)   # ... and half of 0.5 below 2


def test_mx4_pack_rejects_non_finite():
    lib = _lib.load()
    w = np.zeros((32, 3, 32), dtype=np.float32)
    w[3, 1, 2] = np.nan
    out = np.zeros(lib.mi355_packed_conv_weight_mx_bytes(32, 3, 32), dtype=np.uint8)
    assert lib.mi355_pack_conv_weight_mx4_host(w.ctypes.data, 32, 3, 32, out.ctypes.data) != 0
    assert b"non-finite" in lib.mi355_last_error()
