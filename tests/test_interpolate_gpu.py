"""``mlx_audio_amd.tts.models.interpolate`` (HIP ``mi355_interpolate1d``) on the reference's own vectors
(``mlx_audio/tts/tests/test_interpolate.py:40-97``, committed in tests/golden/reference_vectors.json) and bit-exact against the numpy oracle
(oracle/interp_ref.py) on seeded inputs: the op is gathers plus one fp32 blend whose roundings are all pinned, so the bar is 0 ulp."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import interp_ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def interp():
    from mlx_audio_amd import ops
    from mlx_audio_amd.tts.models import interpolate as m

    ops.require_gpu()
    return m


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        return json.load(f)["interpolate"]


def dev(a):
    return torch.from_numpy(np.asarray(a, np.float32)).cuda()


def test_nearest_upsample_and_downsample(interp, golden):
    x = dev([[[1.0, 2.0, 3.0, 4.0]]])
    np.testing.assert_array_equal(interp.interpolate(x, size=8, mode="nearest").cpu().numpy()[0, 0], np.float32(golden["nearest_up8"]))
    np.testing.assert_array_equal(interp.interpolate(x, size=2, mode="nearest").cpu().numpy()[0, 0], np.float32(golden["nearest_down2"]))
    assert interp.interpolate(x, scale_factor=2.0, mode="nearest").shape == (1, 1, 8)


def test_linear_align_corners_vectors(interp, golden):
    y = dev([[[1.0, 3.0, 5.0, 7.0]]])
    t = interp.interpolate(y, size=7, mode="linear", align_corners=True).cpu().numpy()[0, 0]
    f = interp.interpolate(y, size=7, mode="linear", align_corners=False).cpu().numpy()[0, 0]
    np.testing.assert_allclose(t, golden["linear_ac_true_7"], rtol=golden["rtol"])
    np.testing.assert_allclose(f, golden["linear_ac_false_7"], rtol=golden["rtol"])
    d = interp.interpolate(y, size=7, mode="linear").cpu().numpy()[0, 0]  # default = align_corners False (interpolate.py:98)
    np.testing.assert_array_equal(d, f)


def test_linear_width_one_broadcasts(interp):
    one = dev([[[5.0]]])
    np.testing.assert_array_equal(interp.interpolate(one, size=4, mode="linear").cpu().numpy()[0, 0], np.float32([5.0] * 4))
    np.testing.assert_array_equal(interp.interpolate(one, size=3, mode="nearest").cpu().numpy()[0, 0], np.float32([5.0] * 3))


def test_validation_errors(interp):
    x = torch.zeros(1, 1, 4, device="cuda")
    with pytest.raises(ValueError, match="at least 3D"):
        interp.interpolate(torch.zeros(4, 4, device="cuda"), size=8)
    with pytest.raises(ValueError, match="Only one of size or scale_factor"):
        interp.interpolate(x, size=8, scale_factor=2.0)
    with pytest.raises(ValueError, match="One of size or scale_factor"):
        interp.interpolate(x)
    with pytest.raises(ValueError, match="Only 1D interpolation"):
        interp.interpolate(torch.zeros(1, 1, 4, 4, device="cuda"), size=8)


@pytest.mark.parametrize("mode,align", [("nearest", None), ("linear", False), ("linear", True)])
@pytest.mark.parametrize("n,c,w,size", [(2, 3, 17, 40), (1, 5, 129, 33), (3, 1, 1000, 300000), (1, 2, 2, 1), (2, 9, 400, 400)])
def test_bit_exact_vs_oracle(interp, mode, align, n, c, w, size):
    rng = np.random.default_rng(1000 * w + size)
    x = rng.standard_normal((n, c, w)).astype(np.float32)
    ref = interp_ref.interpolate1d(x, size, mode, align)
    got = interp.interpolate1d(dev(x), size, mode, align).cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)


def test_scale_factor_size_rule(interp):
    # interpolate.py:46-54: size = max(1, ceil(W * scale_factor))
    x = torch.zeros(2, 3, 10, device="cuda")
    assert interp.interpolate(x, scale_factor=0.25).shape == (2, 3, 3)
    assert interp.interpolate(x, scale_factor=1 / 300).shape == (2, 3, 1)
    assert interp.interpolate(x, scale_factor=[300.0], mode="linear").shape == (2, 3, 3000)
