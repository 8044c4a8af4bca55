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
