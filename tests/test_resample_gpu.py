"""``mi355_resample_poly`` (the reference's polyphase FIR, mlx_audio/resample.py:15-47, on the GPU) against the reference's own host call,
``scipy.signal.resample_poly(x, up, down, window=kaiser_best taps, padtype="edge")`` as restated in ``mlx_audio_amd.resample.resample_audio_array``
(float64 taps and sums, float32 out).  The kernel sums in float64 too: the bar is one float32 rounding (1.2e-7 of the peak), i.e. the last bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BAR = 1.2e-7


def _host(x, orig, target, axis=-1):
    from mlx_audio_amd.resample import resample_audio_array

    return resample_audio_array(x, orig, target, axis=axis)


@pytest.mark.parametrize("orig,target,n", [(24000, 16000, 24000), (16000, 24000, 7777), (44100, 16000, 44100), (22050, 24000, 30001), (48000, 16000, 501),
                                           (8000, 16000, 64), (16000, 8000, 3), (24000, 16000, 1), (48000, 8000, 100000), (8000, 48000, 999), (96000, 8000, 50000),
                                           (128000, 8000, 64000)])   # the last one takes the one-output-per-thread kernel (its 4-output window exceeds LDS)
def test_resample_kernel_equals_the_host_polyphase_filter(orig, target, n):
    from mlx_audio_amd.resample import resample_on_device

    x = (np.random.default_rng(n).standard_normal(n) * 0.3 + 0.5).astype(np.float32)
    want = _host(x, orig, target)
    got = resample_on_device(torch.from_numpy(x).cuda(), orig, target)
    assert got.is_cuda and got.dtype == torch.float32 and tuple(got.shape) == want.shape
    err = np.abs(got.cpu().numpy() - want).max()
    assert err <= BAR * max(1.0, float(np.abs(want).max())), err


def test_resample_rows_axes_and_the_dispatch_of_resample_audio():
    """Several channels at once, time on either axis, non-contiguous input; ``utils.resample_audio`` (utils.py:541-578) keeps a CUDA tensor on the GPU and
    returns the input itself at equal rates."""
    from mlx_audio_amd.utils import resample_audio

    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 2, 5000)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    got = resample_audio(xd, 24000, 16000)
    assert got.is_cuda and np.abs(got.cpu().numpy() - _host(x, 24000, 16000)).max() <= 4 * BAR * 5
    tf = np.ascontiguousarray(x[0].T)                        # time-first [5000, 2]
    got = resample_audio(torch.from_numpy(tf).cuda(), 24000, 16000, axis=0)
    assert tuple(got.shape) == (3334, 2) and np.abs(got.cpu().numpy() - _host(tf, 24000, 16000, axis=0)).max() <= 4 * BAR * 5
    strided = xd[:, :, ::2]                                   # a view with a stride on the time axis
    assert np.abs(resample_audio(strided, 12000, 16000).cpu().numpy() - _host(x[:, :, ::2], 12000, 16000)).max() <= 4 * BAR * 5
    assert resample_audio(xd, 16000, 16000) is xd
    half = resample_audio(xd.to(torch.float16), 24000, 16000)
    assert half.dtype == torch.float32 and np.abs(half.cpu().numpy() - _host(x.astype(np.float16).astype(np.float32), 24000, 16000)).max() <= 4 * BAR * 5


def test_resample_properties_at_full_size_and_loud_refusal():
    """30 s of 44.1 kHz audio (the Whisper front door): length, linearity (the kernel is a linear map: a*x + b*y), a constant stays that constant through
    the edge padding, energy above the new Nyquist is removed (test_dsp.py:299-349).  A conversion whose input window does not fit LDS is refused."""
    from mlx_audio_amd import ops
    from mlx_audio_amd.resample import polyphase_table, resample_on_device

    n = 30 * 44100
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(n, device="cuda", generator=g)
    b = torch.randn(n, device="cuda", generator=g)
    ya, yb = resample_on_device(a, 44100, 16000), resample_on_device(b, 44100, 16000)
    assert ya.numel() == 480000
    yab = resample_on_device(0.5 * a - 2.0 * b, 44100, 16000)
    assert (yab - (0.5 * ya - 2.0 * yb)).abs().max().item() <= 5e-6
    const = resample_on_device(torch.full((n,), 0.25, device="cuda"), 44100, 16000)
    assert (const - 0.25).abs().max().item() <= 2e-5          # the kaiser_best pass-band ripple at DC
    t = torch.arange(n, device="cuda", dtype=torch.float64) / 44100
    hi = resample_on_device(torch.sin(2 * np.pi * 12000.0 * t).float(), 44100, 16000)   # above the 8 kHz Nyquist of the target
    lo = resample_on_device(torch.sin(2 * np.pi * 1000.0 * t).float(), 44100, 16000)
    assert hi[2000:-2000].abs().max().item() < 1e-3 and abs(lo[2000:-2000].abs().max().item() - 1.0) < 1e-3
    up, down, table, first, n_out = polyphase_table(384000, 8000, 4800)
    with pytest.raises(RuntimeError, match="LDS"):
        ops.resample_poly(torch.zeros(1, 4800, device="cuda"), torch.from_numpy(table).cuda(), up, down, first, n_out)
