"""BigVGAN on the HIP path: the anti-aliased activation kernel and the whole vocoder against the CPU oracle and against the
reference's own modules' output (tests/golden/ref_bigvgan_tiny.npz); shape pins of the reference's tests (codec/tests/test_bigvgan.py).  Needs an MI355X."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def snr_db(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300)))


@pytest.mark.parametrize("B,L,C", [(2, 150, 64), (1, 64, 96), (3, 37, 20), (1, 1, 8), (2, 129, 256)])
def test_anti_aliased_activation_kernel(B, L, C):
    """``mi355_aa_activation`` vs down(act(up(x))) restated with torch convs (oracle/bigvgan_ref.py): every length class (one tile, tile boundary, a single
    sample), ragged utterances, channel counts off the 64-lane tile."""
    from mlx_audio_amd import ops
    from oracle import bigvgan_ref as R

    g = torch.Generator().manual_seed(L + C)
    x = torch.randn(B, L, C, generator=g) * 1.5
    alpha = torch.exp(torch.randn(C, generator=g) * 0.3)
    beta = torch.exp(torch.randn(C, generator=g) * 0.3)
    filt = R.kaiser_sinc_filter1d(0.25, 0.3, 12)
    lens = torch.tensor([L, max(1, L // 2), max(1, L - 1)][:B], dtype=torch.int32)
    y = torch.full((B, L, C + 8), 7.0, device=DEV)[:, :, :C]
    ops.aa_activation(x.to(DEV), y, filt.to(DEV), filt.to(DEV), alpha.to(DEV), (1.0 / (beta + 1e-9)).to(DEV), lens=lens.to(DEV) if B > 1 else None)
    torch.cuda.synchronize()
    for b in range(B):
        n = int(lens[b]) if B > 1 else L
        u = R.upsample2(x[b:b + 1, :n], filt)
        want = R.downsample2(u + (1.0 / (beta + 1e-9)) * torch.sin(u * alpha) ** 2, filt)[0]
        got = y[b, :n].cpu()
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (b, float((got - want).abs().max()))
        assert bool((y[b, n:] == 7.0).all())  # rows past the utterance are untouched


def _pair(cfg, seed):
    from mlx_audio_amd.codec.models.bigvgan import BigVGAN, BigVGANConfig, make_bigvgan_weights
    from oracle.bigvgan_ref import BigVGANRef

    w = make_bigvgan_weights(BigVGANConfig(**cfg), seed=seed)
    return BigVGAN(BigVGANConfig(**cfg), weights=w, device=DEV), BigVGANRef(w, cfg)


@pytest.mark.parametrize("kind", ["1", "2"])
def test_stages_and_waveform_vs_oracle_and_reference_run(kind):
    """float32 checkpoint held as fp16 images (precision 4): every stage within 2e-3 of its peak against the float32 oracle; the waveform within
    SNR >= 50 dB and 2e-3 of the peak against the oracle AND against the reference's own run of the same checkpoint (ref_bigvgan_tiny.npz)."""
    fx = np.load(os.path.join(GOLD, "ref_bigvgan_tiny.npz"))
    cfg = dict(json.loads(str(fx["config"])), resblock=kind)
    eng, ref = _pair(cfg, int(fx["seed_w"]))
    mel = torch.from_numpy((np.random.default_rng(int(fx["seed_mel"])).standard_normal((2, cfg["num_mels"], int(fx["n_frames"]))) * 0.8).astype(np.float32))
    want, wst = ref(mel, return_stages=True)
    got, gst = eng(mel, return_stages=True)
    torch.cuda.synchronize()
    assert tuple(got.shape) == tuple(want.shape) == (2, 1, int(fx["n_frames"]) * math.prod(cfg["upsample_rates"]))
    errs = {k: float((gst[k].cpu() - wst[k]).abs().max() / wst[k].abs().max()) for k in wst}
    g, w_, r = got.cpu().numpy(), want.numpy(), fx[f"audio{kind}"]
    peak = float(np.abs(r).max())
    print(f"bigvgan resblock {kind}: stage rel err {errs}; waveform vs oracle max-abs {np.abs(g - w_).max():.2e} snr {snr_db(g, w_):.1f} dB; "
          f"vs reference run max-abs {np.abs(g - r).max():.2e} snr {snr_db(g, r):.1f} dB (peak {peak:.3f})")
    assert max(errs.values()) < 2e-3
    for tgt in (w_, r):
        assert float(np.abs(g - tgt).max()) <= 2e-3 * peak and snr_db(g, tgt) >= 50.0
    one = eng(mel[:1])  # a batch equals its items
    torch.cuda.synchronize()
    assert float((one - got[:1]).abs().max()) <= 1e-5


def test_reference_shape_pins_and_errors():
    """codec/tests/test_bigvgan.py:10-49 at reduced width (lengths do not depend on it): x256 with tanh, x512 with the clip head."""
    from mlx_audio_amd.codec.models.bigvgan import BigVGAN, BigVGANConfig

    for mels, rates, kers, tanh, bias in ((80, [4, 4, 2, 2, 2, 2], [8, 8, 4, 4, 4, 4], True, True), (128, [8, 4, 2, 2, 2, 2], [16, 8, 4, 4, 4, 4], False, False)):
        cfg = BigVGANConfig(num_mels=mels, upsample_rates=rates, upsample_kernel_sizes=kers, upsample_initial_channel=128, resblock="1",
                            resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, activation="snakebeta", snake_logscale=True,
                            use_bias_at_final=bias, use_tanh_at_final=tanh)
        y = BigVGAN(cfg, device=DEV)(torch.zeros(1, mels, 100))
        torch.cuda.synchronize()
        assert tuple(y.shape) == (1, 1, 100 * math.prod(rates)) and bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
    with pytest.raises(NotImplementedError):
        BigVGAN(BigVGANConfig(num_mels=8, upsample_rates=[2], upsample_kernel_sizes=[4], upsample_initial_channel=16, resblock="1", resblock_kernel_sizes=[3],
                              resblock_dilation_sizes=[[1]], activation="snake", snake_logscale=True), device=DEV)
    eng = BigVGAN(cfg, device=DEV)
    with pytest.raises(ValueError):
        eng(torch.zeros(1, 7, 10))
    # sanitize: PyTorch layouts -> the layouts this model allocates; batch-norm bookkeeping dropped
    from mlx_audio_amd.codec.models.bigvgan import make_bigvgan_weights

    w = make_bigvgan_weights(cfg, seed=1)
    pt = {}
    for k, v in w.items():
        if "ups." in k and v.dim() == 3 and k.endswith("weight_v"):
            pt[k] = v.permute(2, 0, 1).contiguous()      # (out, K, in) -> PyTorch ConvTranspose1d (in, out, K)
        elif v.dim() == 3 and ("conv" in k or "filter" in k) and k.endswith(("weight_v", "filter")):
            pt[k] = v.permute(0, 2, 1).contiguous()      # (out, K, in) -> (out, in, K)
        else:
            pt[k] = v
    pt["conv_pre.num_batches_tracked"] = torch.zeros(1)
    out = eng.sanitize(pt)
    assert set(out) == set(w) and all(torch.equal(out[k], w[k]) for k in w)
