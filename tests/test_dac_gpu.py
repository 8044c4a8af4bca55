"""Descript Audio Codec decode on the HIP path vs the CPU oracle; length pins of the reference's own tests
(codec/tests/test_descript.py:41-42, 107-108: 250 frames -> 80 043 samples, 430 -> 220 235).  Needs a real MI355X: ``pytest -m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_peak(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def snr_db(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float(10 * torch.log10(ref.pow(2).sum() / ((got - ref).pow(2).sum() + 1e-30)))


def _pair(rates, dim, latent, nq, seed, fp16_exact, csize=1024):
    from mlx_audio_amd.codec.models.descript import DAC, make_dac_weights
    from oracle.dac_ref import DACDecoderRef, wn_conv_weight, wn_convT_weight

    w = make_dac_weights(dim, rates, latent, nq, csize, 8, seed=seed)
    if fp16_exact:  # make the FOLDED conv weights fp16-representable (g := 1-norm trick: v := folded weight rounded, g := its norm)
        for k in [k for k in w if k.endswith("weight_v") and k.startswith("decoder.")]:
            base = k[: -len(".weight_v")]
            tr = ".block.layers.1" in base and base.count(".block.layers.") == 1
            folded = (wn_convT_weight if tr else wn_conv_weight)(w[base + ".weight_g"], w[k]).half().float()
            dims = (0, 1) if tr else (1, 2)
            w[base + ".weight_g"] = torch.sqrt((folded.double() ** 2).sum(dim=dims, keepdim=True)).float()
            w[k] = folded
    eng = DAC(decoder_dim=dim, decoder_rates=rates, latent_dim=latent, n_codebooks=nq, codebook_size=csize, codebook_dim=8, sample_rate=16000,
              weights=w, device=DEV)
    return eng, DACDecoderRef(w, rates, nq)


def test_from_codes_and_decode_stages_vs_oracle():
    """Small widths, all four rates (even and odd strides): every stage and the waveform vs the float32 oracle.  With fp16-exact folded
    weights the path is exact to the activation split (3e-5); with float32 weights the fp16 weight image is the stated deviation."""
    for exact in (True, False):
        eng, ref = _pair([8, 5, 4, 2], 256, 64, 4, 11, exact)
        g = torch.Generator().manual_seed(5)
        codes = torch.randint(0, 1024, (2, 4, 37), generator=g)
        z_ref = ref.from_codes(codes)
        z, z_p, c = eng.quantizer.from_codes(codes)
        torch.cuda.synchronize()
        assert tuple(z.shape) == (2, 64, 37) and tuple(z_p.shape) == (2, 32, 37) and c is codes
        assert rel_peak(z, z_ref) < 2e-6
        want, wst = ref.decode(z_ref, return_stages=True)
        got, gst = eng.decode(z, return_stages=True)
        torch.cuda.synchronize()
        assert tuple(got.shape) == tuple(want.shape)
        errs = {k: rel_peak(gst[k], wst[k]) for k in wst}
        s = snr_db(got, want)
        err = float((got.cpu() - want).abs().max())
        print(f"dac exact_fp16_weights={exact}: stage rel err {errs} waveform max_abs={err:.2e} snr={s:.1f} dB")
        if exact:
            assert max(errs.values()) < 5e-5 and s > 85.0 and err < 1e-4, (errs, s, err)
        else:
            assert max(errs.values()) < 2e-3 and s >= 50.0 and err <= 2e-3, (errs, s, err)
        # a batch equals its items
        one = eng.decode(z[:1])
        assert snr_db(one[0], got[0]) > 100.0


def test_reference_length_pins_full_width():
    """The 16 kHz model of the reference test (decoder_dim 1536, rates 8 / 5 / 4 / 2, 12 codebooks): 250 frames -> (1, 80043, 1), and the
    tail sample produced by the groups-as-output_padding slip matches the oracle's."""
    eng, ref = _pair([8, 5, 4, 2], 1536, 1024, 12, 2, True)
    codes = torch.randint(0, 1024, (1, 12, 250), generator=torch.Generator().manual_seed(1))
    z, _, _ = eng.quantizer.from_codes(codes)
    y = eng.decode(z)
    torch.cuda.synchronize()
    assert tuple(z.shape) == (1, 1024, 250) and tuple(y.shape) == (1, 80_043, 1) and torch.isfinite(y).all() and float(y.abs().max()) <= 1.0
    zs = z[:, :, :20]
    want = ref.decode(ref.from_codes(codes[:, :, :20]))
    got = eng.decode(zs)
    assert tuple(got.shape) == tuple(want.shape) == (1, 20 * 320 + 43, 1)
    assert snr_db(got, want) > 80.0 and float((got.cpu() - want).abs()[:, -64:].max()) < 1e-4
    # 44 kHz rates: all-even strides
    from mlx_audio_amd.codec.models.descript import DAC
    eng44 = DAC(decoder_dim=96, decoder_rates=[8, 8, 4, 2], latent_dim=32, n_codebooks=2, codebook_size=64, sample_rate=44100, device=DEV)
    assert tuple(eng44.decode(torch.zeros(1, 32, 430)).shape) == (1, 220_235, 1)


def test_errors_are_loud():
    eng, _ = _pair([4, 2], 64, 32, 2, 0, False, csize=64)
    with pytest.raises(IndexError):
        eng.quantizer.from_codes(torch.full((1, 2, 5), 64))
    with pytest.raises(IndexError):
        eng.quantizer.from_codes(torch.zeros((1, 3, 5), dtype=torch.long))
    with pytest.raises(ValueError, match="decode-only"):   # this engine was loaded without encoder weights (tests/test_codec_encode_gpu.py has the encode side)
        eng.encode(torch.zeros(1, 1, 800))
    with pytest.raises(ValueError, match="decode-only"):
        eng(torch.zeros(1, 1, 800))
    assert tuple(eng.preprocess(torch.zeros(1, 1, 803), 16000).shape) == (1, 1, 960)   # right-pad to a multiple of the encoder hop 320 (dac.py:180-187)
    with pytest.raises(AssertionError):
        eng.preprocess(torch.zeros(1, 1, 800), 8000)
