"""SNAC decode on the HIP path vs the CPU oracle; the length pin of the reference's own test
(codec/tests/test_snac.py:24-34: 59 / 118 / 236 code frames -> 120 907 samples).  Needs a real MI355X: ``pytest -m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CFG_24K = dict(sampling_rate=24000, encoder_dim=48, encoder_rates=[2, 4, 8, 8], decoder_dim=1024, decoder_rates=[8, 8, 4, 2], attn_window_size=None,
               codebook_size=4096, codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True)   # the reference test's config


def rel_peak(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def snr_db(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float(10 * torch.log10(ref.pow(2).sum() / ((got - ref).pow(2).sum() + 1e-30)))


def _pair(cfg, seed, fp16_exact):
    from mlx_audio_amd.codec.models.snac import SNAC, make_snac_weights
    from oracle.snac_ref import SNACDecoderRef, wn_weight

    latent = cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
    w = make_snac_weights(latent, cfg["decoder_dim"], cfg["decoder_rates"], cfg["vq_strides"], cfg["codebook_size"], cfg["codebook_dim"], cfg["noise"],
                          cfg["depthwise"], seed=seed, attn=cfg["attn_window_size"] is not None)
    if fp16_exact:  # make the FOLDED conv weights fp16-representable: v := folded weight rounded, g := its norm
        for k in [k for k in w if k.endswith("weight_v") and k.startswith("decoder.")]:
            base = k[: -len(".weight_v")]
            folded = wn_weight(w[base + ".weight_g"], w[k]).half().float()
            w[base + ".weight_g"] = torch.sqrt((folded.double() ** 2).sum(dim=(1, 2), keepdim=True)).float()
            w[k] = folded
    return SNAC(**cfg, weights=w, device=DEV), SNACDecoderRef(w, cfg["decoder_rates"], cfg["vq_strides"], cfg["noise"], cfg["depthwise"],
                                                              attn_window_size=cfg["attn_window_size"])


def _codes(cfg, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, cfg["codebook_size"], (B, T // s), generator=g) for s in cfg["vq_strides"]]


@pytest.mark.parametrize("depthwise,noise", [(True, True), (False, True), (True, False)])
def test_from_codes_and_decode_stages_vs_oracle(depthwise, noise):
    """Small widths, three code levels with strides 4 / 2 / 1, explicit NoiseBlock noise: every stage and the waveform vs the float32 oracle.
    With fp16-exact folded weights the path is exact to the activation split; with float32 weights the fp16 weight image is the deviation."""
    cfg = dict(sampling_rate=24000, encoder_dim=4, encoder_rates=[2, 4, 8, 8], decoder_dim=256, decoder_rates=[8, 5, 4, 2], attn_window_size=None,
               codebook_size=512, codebook_dim=8, vq_strides=[4, 2, 1], noise=noise, depthwise=depthwise)
    for exact in (True, False):
        eng, ref = _pair(cfg, 11, exact)
        codes = _codes(cfg, 2, 36, 5)
        z_ref = ref.from_codes(codes)
        z = eng.quantizer.from_codes(codes)
        torch.cuda.synchronize()
        assert tuple(z.shape) == (2, 64, 36)
        assert rel_peak(z, z_ref) < 2e-6
        g = torch.Generator().manual_seed(9)
        lens, L = [], 36
        for s in cfg["decoder_rates"]:
            L = (L - 1) * s - 2 * ((s + 1) // 2) + 2 * s + 1
            lens.append(L)
        # NoiseBlock noise is [B, 1, channels] (the reference's quirk: oracle/snac_ref.py); the block widths halve from decoder_dim
        noises = [torch.randn(2, 1, cfg["decoder_dim"] >> (i + 1), generator=g) for i in range(len(lens))]
        want, wst = ref.decode(z_ref, noises, return_stages=True)
        got, gst = eng.decode_latents(z, noises, return_stages=True)
        torch.cuda.synchronize()
        assert tuple(got.shape) == tuple(want.shape) == (2, lens[-1], 1)
        errs = {k: rel_peak(gst[k], wst[k]) for k in wst}
        s = snr_db(got, want)
        err = float((got.cpu() - want).abs().max())
        print(f"snac depthwise={depthwise} noise={noise} exact_fp16_weights={exact}: stage rel err {errs} waveform max_abs={err:.2e} snr={s:.1f} dB")
        if exact:
            assert max(errs.values()) < 5e-5 and s > 85.0 and err < 1e-4, (errs, s, err)
        else:
            assert max(errs.values()) < 2e-3 and s >= 50.0 and err <= 2e-3, (errs, s, err)
        one = eng.decode_latents(z[:1], [n[:1] for n in noises])   # a batch equals its items
        assert snr_db(one[0], got[0]) > 100.0
        full = eng.decode(codes, noises)                            # decode(codes) = decoder(from_codes(codes))
        assert torch.equal(full, got)


def test_local_mha_variant_vs_oracle():
    """The 32 / 44 kHz models' ``LocalMHA`` (attention.py:5-53: LayerNorm, to_qkv, windows of 32 positions, rotate-half rotary embedding inside the window,
    softmax, to_out, + x) between the input convs and the first decoder block.  PARITY UNPINNED: the reference's own module raises in its decoder
    (tests/golden/ref_snac_local_mha_probe.json), so this is held to the oracle's restatement of what the module means."""
    cfg = dict(sampling_rate=32000, encoder_dim=4, encoder_rates=[2, 4, 8, 8], decoder_dim=256, decoder_rates=[8, 5, 4, 2], attn_window_size=32,
               codebook_size=512, codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True)
    eng, ref = _pair(cfg, 5, False)
    codes = _codes(cfg, 2, 64, 7)   # 64 positions = two windows
    z_ref = ref.from_codes(codes)
    z = eng.quantizer.from_codes(codes)
    g = torch.Generator().manual_seed(2)
    noises = [torch.randn(2, 1, cfg["decoder_dim"] >> (i + 1), generator=g) for i in range(4)]
    want, wst = ref.decode(z_ref, noises, return_stages=True)
    got, gst = eng.decode_latents(z, noises, return_stages=True)
    torch.cuda.synchronize()
    assert "attn" in gst and rel_peak(gst["attn"], wst["attn"]) < 1e-3, rel_peak(gst["attn"], wst["attn"])
    assert rel_peak(gst["attn"], wst["conv_in"]) > 1e-2          # the block is not a no-op
    assert snr_db(got, want) >= 50.0 and float((got.cpu() - want).abs().max()) <= 2e-3 * float(want.abs().max())


def test_reference_length_pin_full_width():
    """The 24 kHz model of the reference test: codes of 59 / 118 / 236 frames -> (1, 120907, 1); a short prefix against the oracle, including
    the tail samples produced by the groups-as-output_padding slip."""
    eng, ref = _pair(CFG_24K, 2, True)
    codes = _codes(CFG_24K, 1, 236, 1)
    assert [tuple(c.shape) for c in codes] == [(1, 59), (1, 118), (1, 236)]
    y = eng.decode(codes)   # NoiseBlock noise drawn on the device
    torch.cuda.synchronize()
    assert tuple(y.shape) == (1, 120_907, 1) and torch.isfinite(y).all() and float(y.abs().max()) <= 1.0
    short = [c[:, : 16 // s] for c, s in zip(codes, CFG_24K["vq_strides"])]
    lens, L = [], 16
    for s in CFG_24K["decoder_rates"]:
        L = L * s + 1
        lens.append(L)
    g = torch.Generator().manual_seed(3)
    noises = [torch.randn(1, 1, CFG_24K["decoder_dim"] >> (i + 1), generator=g) for i in range(len(lens))]
    want = ref.decode(ref.from_codes(short), noises)
    got = eng.decode(short, noises)
    assert tuple(got.shape) == tuple(want.shape) == (1, lens[-1], 1)
    assert snr_db(got, want) > 80.0 and float((got.cpu() - want).abs()[:, -64:].max()) < 1e-4


def test_surface_and_errors_are_loud():
    from mlx_audio_amd.codec.models.snac import SNAC

    cfg = dict(sampling_rate=24000, encoder_dim=2, encoder_rates=[2, 4, 8, 8], decoder_dim=64, decoder_rates=[4, 2], attn_window_size=None,
               codebook_size=64, codebook_dim=8, vq_strides=[2, 1], noise=True, depthwise=True)
    from mlx_audio_amd.codec.models.snac import make_snac_weights

    dec_only = lambda attn=False: make_snac_weights(32, 64, [4, 2], [2, 1], 64, 8, True, True, seed=0, attn=attn)   # noqa: E731  (no encoder.* / in_proj keys)
    eng = SNAC(**cfg, weights=dec_only(), device=DEV)
    assert eng.hop_length == 512 and eng.latent_dim == 32 and eng.n_codebooks == 2
    with pytest.raises(IndexError):
        eng.quantizer.from_codes([torch.full((1, 2), 64), torch.zeros((1, 4), dtype=torch.long)])
    with pytest.raises(IndexError):
        eng.quantizer.from_codes([torch.zeros((1, 4), dtype=torch.long)])
    with pytest.raises(ValueError):
        eng.quantizer.from_codes([torch.zeros((1, 2), dtype=torch.long), torch.zeros((1, 5), dtype=torch.long)])
    with pytest.raises(ValueError, match="decode-only"):   # loaded without encoder weights (tests/test_codec_encode_gpu.py has the encode side)
        eng.encode(torch.zeros(1, 1, 800))
    with pytest.raises(ValueError, match="decode-only"):
        eng(torch.zeros(1, 1, 800))
    with pytest.raises(ValueError):   # LocalMHA: positions must be a whole number of windows
        SNAC(**{**cfg, "attn_window_size": 32}, weights=dec_only(attn=True), device=DEV).decode([torch.zeros((1, 2), dtype=torch.long), torch.zeros((1, 4), dtype=torch.long)])
    assert tuple(eng.preprocess(torch.zeros(1, 1, 1000)).shape) == (1, 1, 1024)   # right-pad to hop 512 * lcm(2, 1) (snac.py:67-86)
    # decode_stream (snac.py:109-165): first call decodes as is and keeps the last context_frames codes per level
    codes = [torch.zeros((1, 6), dtype=torch.long), torch.zeros((1, 12), dtype=torch.long)]
    audio, ctx = eng.decode_stream(codes, None, context_frames=4)
    assert tuple(audio.shape) == (1, ((12 * 4 + 1) * 2 + 1), 1) and [tuple(c.shape) for c in ctx] == [(1, 4), (1, 4)]
    audio2, ctx2 = eng.decode_stream(codes, ctx, context_frames=4)
    assert audio2.shape[1] == ((12 + 4) * 4 + 1) * 2 + 1 and [tuple(c.shape) for c in ctx2] == [(1, 4), (1, 4)]
