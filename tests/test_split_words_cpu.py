"""SPLIT words (include/mi355audio.h: x_split / y_split / mi355_split16): the numpy statement of the format (oracle/mx_ref.py) against hand-worked cases and
its own invariants.  The device kernels are compared with it in tests/test_conv_split_gpu.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mx_ref  # noqa: E402


def test_split_words_hand_cases():
    # 1 + 2^-12 in IEEE half: hi = 1.0 (0x3c00), lo = 2^-12 (0x0c00)
    w = mx_ref.split16_words(np.array([1.0 + 2.0 ** -12], np.float32), 4)
    assert int(w[0]) == (0x0c00 << 16) | 0x3c00
    # bfloat16: 1 + 2^-9 -> hi = 1.0 (0x3f80), lo = 2^-9 (0x3b00)
    w = mx_ref.split16_words(np.array([1.0 + 2.0 ** -9], np.float32), 2)
    assert int(w[0]) == (0x3b00 << 16) | 0x3f80
    # ties to even in the hi part: 1 + 2^-11 (half way between 1 and 1 + 2^-10) -> hi = 1.0, lo = +2^-11
    w = mx_ref.split16_words(np.array([1.0 + 2.0 ** -11], np.float32), 4)
    assert int(w[0]) & 0xffff == 0x3c00 and int(w[0]) >> 16 == 0x1000
    # beyond the half range: clamped, finite, lo = 0
    w = mx_ref.split16_words(np.array([1.0e6, -1.0e6], np.float32), 4)
    assert [int(x) for x in w] == [0x7bff, 0xfbff]
    assert int(mx_ref.split16_words(np.array([0.0], np.float32), 4)[0]) == 0


def test_split_words_carry_22_bits_half_and_16_bits_bf16():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-8, 8, 200000))).astype(np.float32)
    for fmt, rel in ((4, 2.0 ** -21), (2, 2.0 ** -15)):
        v = mx_ref.split16_value(mx_ref.split16_words(x, fmt), fmt)
        err = np.abs(v - x.astype(np.float64))
        # (IEEE half: a lo part under 2^-14 lands on the subnormal quantum 2^-24 -- an absolute floor of 2^-25 whatever the value)
        floor = 2.0 ** -25 if fmt == 4 else 0.0
        assert np.all(err <= rel * np.abs(x) + floor), (fmt, float((err / np.abs(x)).max()))


def test_split_words_hi_is_the_rounded_value():
    rng = np.random.default_rng(8)
    x = rng.standard_normal(4096).astype(np.float32) * 3
    w4 = mx_ref.split16_words(x, 4)
    assert np.array_equal((w4 & 0xffff).astype(np.uint16), x.astype(np.float16).view(np.uint16))
    w2 = mx_ref.split16_words(x, 2)
    import torch
    ref = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal((w2 & 0xffff).astype(np.uint16), ref)
