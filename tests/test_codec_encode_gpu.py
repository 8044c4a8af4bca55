"""Codec ENCODE sides on the HIP path: DAC / SNAC / EnCodec ``encode`` against the CPU oracles (pinned to the
reference's own modules by tests/test_codec_encode_cpu.py) and against the reference-run fixtures themselves.  Float stages within stated bars; codes
bit-exact under the margin rule (tests/_margin.py: a decision whose top-2 gap is at float32 rounding level may fall either way).
Needs a real MI355X: ``pytest -m gpu``."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _margin  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_peak(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_audio(batch, n, sr, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / sr
    rows = [0.5 * torch.sin(2 * np.pi * (170 + 80 * b) * t) * (0.6 + 0.4 * torch.sin(2 * np.pi * 3 * t)) + 0.15 * torch.randn(n, generator=g, dtype=torch.float64)
            for b in range(batch)]
    return torch.stack(rows)[:, None, :].float()


def walk_frames(family, got, want, margins, thr):
    """codes [B, n, T]: the n residual decisions of a frame form one sequence (a different pick changes every later residual of that frame)."""
    got, want, margins = got.cpu(), want.cpu(), margins.cpu()
    for b in range(got.shape[0]):
        for t in range(got.shape[2]):
            _margin.walk(family, got[b, :, t].tolist(), want[b, :, t].tolist(), margins[b, :, t].tolist(), thr=thr, where=(b, t))


def resync_frames(family, run, want, wm, thr, what="", same_input=True):
    """Residual chains [B, n, T] with RE-SYNCHRONISATION (round 6): ``run(force)`` -> (codes, margins) runs the engine's quantizer with the oracle's
    code forced wherever the ORACLE's own top-2 gap is below ``thr`` (its knife edges); every other decision of every frame must then equal the
    oracle's bit for bit (tests/_margin.py: walk_forced).  ``thr`` must cover the build-to-build difference of the gaps: that difference is measured on
    the compared decisions, printed, and asserted to stay below thr."""
    want, wm = torch.as_tensor(want).cpu(), torch.as_tensor(wm).cpu()
    mask = wm < thr
    got, gm = run((mask, want))
    torch.cuda.synchronize()
    got, gm = got.cpu(), gm.cpu()
    assert tuple(got.shape) == tuple(want.shape)
    noise = float((gm - wm)[~mask].abs().max()) if (~mask).any() else 0.0
    print(f"{family} {what}: {int(mask.sum())} of {mask.numel()} decisions forced (oracle gap < {thr:.1e}); gap difference between the builds on the rest: max {noise:.2e}")
    assert noise < thr or not same_input, (family, what, noise, thr)   # (same_input=False: the two runs searched on different latents -- the gaps differ by the latents' error)
    for b in range(got.shape[0]):
        for t in range(got.shape[2]):
            _margin.walk_forced(family, got[b, :, t].tolist(), want[b, :, t].tolist(), mask[b, :, t].tolist(), where=(what, b, t))
    return got, gm


def resync_levels(family, run, want, wm, thr, what=""):
    """The same for SNAC's levels of different rates (lists of [B, T / stride_i])."""
    want = [torch.as_tensor(x).cpu() for x in want]
    wm = [torch.as_tensor(x).cpu() for x in wm]
    masks = [m < thr for m in wm]
    got, gm = run((masks, want))
    torch.cuda.synchronize()
    got, gm = [x.cpu() for x in got], [x.cpu() for x in gm]
    noise = max(float((g - w)[~m].abs().max()) if (~m).any() else 0.0 for g, w, m in zip(gm, wm, masks))
    print(f"{family} {what}: {sum(int(m.sum()) for m in masks)} of {sum(m.numel() for m in masks)} decisions forced (oracle gap < {thr:.1e}); gap difference on the rest: max {noise:.2e}")
    assert noise < thr, (family, what, noise, thr)
    strides = [got[-1].shape[1] // g.shape[1] for g in got]
    for b in range(got[0].shape[0]):
        for t in range(got[-1].shape[1]):
            g = [int(got[i][b, t // s]) for i, s in enumerate(strides)]
            w = [int(want[i][b, t // s]) for i, s in enumerate(strides)]
            f = [bool(masks[i][b, t // s]) for i, s in enumerate(strides)]
            _margin.walk_forced(family, g, w, f, where=(what, b, t))
    return got, gm


# ------------------------------------------------------------------------------------------------------------------ DAC
def fp16_exact_dac(w):
    """Make every FOLDED conv weight fp16-representable (v := the folded weight rounded to half, g := its norm), so the fp16 MFMA image is exact."""
    from oracle.dac_ref import wn_conv_weight, wn_convT_weight

    for k in [k for k in w if k.endswith("weight_v")]:
        base = k[: -len(".weight_v")]
        tr = base.startswith("decoder.") and ".block.layers.1" in base and base.count(".block.layers.") == 1
        folded = (wn_convT_weight if tr else wn_conv_weight)(w[base + ".weight_g"], w[k]).half().float()
        dims = (0, 1) if tr else (1, 2)
        w[base + ".weight_g"] = torch.sqrt((folded.double() ** 2).sum(dim=dims, keepdim=True)).float()
        w[k] = folded
    return w


def dac_pair(c, seed, exact):
    from mlx_audio_amd.codec.models.descript import DAC, make_dac_encoder_weights, make_dac_weights
    from oracle.dac_ref import DACDecoderRef, DACEncoderRef

    w = make_dac_weights(c["decoder_dim"], c["decoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_size"], c["codebook_dim"], seed=seed)
    w.update(make_dac_encoder_weights(c["encoder_dim"], c["encoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_dim"], seed=seed))
    if exact:
        fp16_exact_dac(w)
    eng = DAC(**c, weights=w, device=DEV)
    return eng, DACEncoderRef(w, c["encoder_rates"], c["n_codebooks"]), DACDecoderRef(w, c["decoder_rates"], c["n_codebooks"])


def test_dac_encode_against_the_reference_run():
    """The reference's own ``DAC.encode`` / ``DAC.__call__`` outputs (tests/golden/ref_dac_encode.npz): encoder latents, codes of every codebook under the
    margin rule, latents, z_q, losses, ``n_quantizers = 2``, and the ``__call__`` round trip."""
    from mlx_audio_amd.codec.models.descript import DAC
    from test_codec_encode_cpu import dac_model_weights
    from oracle.dac_ref import DACEncoderRef

    fx = np.load(os.path.join(GOLD, "ref_dac_encode.npz"))
    c, w = dac_model_weights(fx)
    eng = DAC(**c, weights=w, device=DEV)
    audio = torch.from_numpy(fx["audio"])
    enc = eng.encoder(audio)
    torch.cuda.synchronize()
    assert tuple(enc.shape) == fx["enc"].shape
    e = rel_peak(enc, fx["enc"])
    print(f"dac encoder vs the reference run: {e:.2e} of the peak (float32 checkpoint held as an fp16 image)")
    assert e < 2e-3, e
    # the search on the REFERENCE's encoder output: only the quantizer differs
    ref = DACEncoderRef(w, c["encoder_rates"], c["n_codebooks"])
    _, _, _, _, _, wm = ref.quantize(torch.from_numpy(fx["enc"]), return_margins=True)
    z, codes, latents, commit, cbl, gm = eng.quantizer(torch.from_numpy(fx["enc"]), return_margins=True)
    torch.cuda.synchronize()
    assert tuple(codes.shape) == fx["codes"].shape and codes.dtype == torch.int64 and tuple(latents.shape) == fx["latents"].shape
    resync_frames("dac_encode", lambda f: (lambda o: (o[1], o[5]))(eng.quantizer(torch.from_numpy(fx["enc"]), return_margins=True, force=f)),
                  torch.from_numpy(fx["codes"]), wm, thr=2e-3, what="reference run")
    same = (codes.cpu().numpy() == fx["codes"]).all(axis=1)            # frames whose whole chain agrees (free-running)
    assert same.mean() > 0.7, same.mean()
    lat = latents.cpu().numpy().reshape(2, c["n_codebooks"], c["codebook_dim"], -1)
    fl = fx["latents"].reshape(2, c["n_codebooks"], c["codebook_dim"], -1)
    for b in range(2):
        for t in np.nonzero(same[b])[0]:
            assert np.abs(lat[b, :, :, t] - fl[b, :, :, t]).max() < 2e-3 * np.abs(fl).max()
    if same.all():
        assert rel_peak(z, fx["z"]) < 1e-5
        assert abs(float(commit) - float(fx["commitment_loss"])) < 2e-3 * float(fx["commitment_loss"]) and float(cbl) == float(commit)
    z2, codes2, latents2, *_ = eng.quantizer(torch.from_numpy(fx["enc"]), 2)
    assert tuple(codes2.shape) == fx["codes_nq2"].shape and tuple(latents2.shape) == fx["latents_nq2"].shape and tuple(z2.shape) == fx["z_nq2"].shape
    assert torch.equal(codes2.cpu(), codes.cpu()[:, :2])
    out = eng(audio, c["sample_rate"])
    torch.cuda.synchronize()
    assert tuple(out["audio"].shape) == fx["call_audio"].shape and tuple(out["codes"].shape) == fx["call_codes"].shape and tuple(out["z"].shape) == fx["call_z"].shape
    assert torch.isfinite(out["audio"]).all()


@pytest.mark.parametrize("exact", [True, False])
def test_dac_encode_stages_and_codes_vs_oracle(exact):
    """Published-shape strides (2 / 4 / 5 / 8: even and odd) at widths that take the wave-specialised kernel, two utterances of 1.3 s whose length is
    not a whole number of hops: every encoder stage, then the codes of the free-running ``encode`` under the margin rule."""
    c = dict(encoder_dim=32, encoder_rates=[2, 4, 5, 8], latent_dim=256, decoder_dim=256, decoder_rates=[8, 5, 4, 2], n_codebooks=6, codebook_size=1024,
             codebook_dim=8, sample_rate=16000)
    eng, ref, dec = dac_pair(c, 17, exact)
    audio = make_audio(2, 320 * 64 + 131, 16000, seed=3)
    zr, est = ref.encoder(audio, return_stages=True)
    z, gst = eng.encoder(audio, return_stages=True)
    torch.cuda.synchronize()
    assert tuple(z.shape) == tuple(zr.shape)
    errs = {k: rel_peak(gst[k], est[k]) for k in est}
    print(f"dac encoder exact_fp16_weights={exact}: stage rel err { {k: f'{v:.1e}' for k, v in errs.items()} }")
    assert max(errs.values()) < (2e-4 if exact else 2e-3), errs
    # exact images: the free-running chain (device latents); float32 checkpoint behind an fp16 image: the search on the ORACLE's latents, so that the
    # knife-edge threshold only has to cover the in_proj image (a latent error of 1e-3 of the peak would put a sixth of all decisions under it)
    want = ref.quantize(zr, return_margins=True)
    got = eng.encode(audio, return_margins=True) if exact else eng.quantizer(zr, return_margins=True)
    torch.cuda.synchronize()
    resync_frames("dac_encode", lambda f: (lambda o: (o[1], o[5]))(eng.encode(audio, return_margins=True, force=f) if exact else eng.quantizer(zr, return_margins=True, force=f)),
                  want[1], want[5], thr=1e-5 if exact else 1.5e-3, what=f"exact={exact}")   # measured gap difference: 1.9e-6 (exact images) / 3.4e-4 (fp16 image of a float32 in_proj)
    agree = float((got[1].cpu() == want[1]).float().mean())
    print(f"dac encode exact_fp16_weights={exact}: {100 * agree:.1f} % of all codes equal the oracle's (free-running)")
    assert agree > 0.85, agree
    # z_q is from_codes of the codes found: the decode side takes it unchanged
    zq2, _, _ = eng.quantizer.from_codes(got[1])
    assert torch.equal(zq2, got[0])
    y = eng.decode(got[0])
    torch.cuda.synchronize()
    assert y.shape[0] == 2 and y.shape[2] == 1 and torch.isfinite(y).all()
    # a batch equals its items (codes of item 1 alone)
    # (the item alone through the same entry point the batch took: the free-running encoder for exact images, the search on the oracle's latents otherwise)
    resync_frames("dac_encode", lambda f: (lambda o: (o[1], o[5]))(eng.encode(audio[1:2], return_margins=True, force=f) if exact else eng.quantizer(zr[1:2], return_margins=True, force=f)),
                  got[1][1:2], got[5][1:2], thr=1e-4, what="item vs batch")


def test_dac_encode_errors():
    from mlx_audio_amd.codec.models.descript import DAC, make_dac_weights

    c = dict(encoder_dim=16, encoder_rates=[2, 4], latent_dim=32, decoder_dim=64, decoder_rates=[4, 2], n_codebooks=2, codebook_size=64, codebook_dim=8, sample_rate=16000)
    dec_only = DAC(**c, weights=make_dac_weights(64, [4, 2], 32, 2, 64, 8, seed=1), device=DEV)
    with pytest.raises(ValueError, match="without encoder weights"):
        dec_only.encode(torch.zeros(1, 1, 800))
    with pytest.raises(ValueError, match="in_proj"):
        dec_only.quantizer(torch.zeros(1, 32, 5))
    eng, _, _ = dac_pair(c, 2, False)
    with pytest.raises(ValueError, match=r"\[B, 1, samples\]"):
        eng.encode(torch.zeros(1, 2, 800))
    with pytest.raises(ValueError, match="fewer than one frame"):
        eng.encode(torch.zeros(1, 1, 1))


# ------------------------------------------------------------------------------------------------------------------ SNAC
def fp16_exact_snac(w):
    """Every FOLDED conv weight fp16-representable (all SNAC weight norms run over every axis but 0, layers.py:9-15)."""
    from oracle.snac_ref import wn_weight

    for k in [k for k in w if k.endswith("weight_v")]:
        base = k[: -len(".weight_v")]
        folded = wn_weight(w[base + ".weight_g"], w[k]).half().float()
        w[base + ".weight_g"] = torch.sqrt((folded.double() ** 2).sum(dim=(1, 2), keepdim=True)).float()
        w[k] = folded
    return w


def walk_levels(family, got, want, gm, wm, thr):
    """SNAC's levels have different rates: the decisions that depend on one another are (level 0 frame t // s0 ... level n frame t): walk them per finest frame."""
    strides = [got[-1].shape[1] // g.shape[1] for g in got]
    for b in range(got[0].shape[0]):
        for t in range(got[-1].shape[1]):
            g = [int(got[i][b, t // s]) for i, s in enumerate(strides)]
            w = [int(want[i][b, t // s]) for i, s in enumerate(strides)]
            m = [min(float(gm[i][b, t // s]), float(wm[i][b, t // s])) for i, s in enumerate(strides)]
            _margin.walk(family, g, w, m, thr=thr, where=(b, t))


@pytest.mark.parametrize("kind", ["dw", "dense"])
def test_snac_encode_against_the_reference_run(kind):
    """The reference's own ``SNAC.encode`` outputs (tests/golden/ref_snac_encode_*.npz): padded length, encoder latents, the codes of the three levels
    under the margin rule (searched on the reference's latents), z_q."""
    from mlx_audio_amd.codec.models.snac import SNAC
    from test_codec_encode_cpu import snac_model_weights
    from oracle.snac_ref import SNACEncoderRef

    fx = np.load(os.path.join(GOLD, f"ref_snac_encode_{kind}.npz"))
    c, w = snac_model_weights(fx)
    eng = SNAC(**c, weights=w, device=DEV)
    audio = torch.from_numpy(fx["audio"])
    padded = eng.preprocess(audio)
    assert padded.shape[-1] == int(fx["padded_len"])
    z = eng.encoder(padded)
    torch.cuda.synchronize()
    e = rel_peak(z, fx["z"])
    print(f"snac ({kind}) encoder vs the reference run: {e:.2e} of the peak")
    assert tuple(z.shape) == fx["z"].shape and e < 2e-3, e
    ref = SNACEncoderRef(w, c["encoder_rates"], c["vq_strides"], depthwise=c["depthwise"])
    _, wc, wm = ref.quantize(torch.from_numpy(fx["z"]), return_margins=True)
    z_q, codes, gm = eng.quantizer(torch.from_numpy(fx["z"]), return_margins=True)
    torch.cuda.synchronize()
    want = [torch.from_numpy(fx[f"codes{i}"]).long() for i in range(len(codes))]
    assert all(tuple(a.shape) == tuple(b.shape) and a.dtype == torch.int64 for a, b in zip(codes, want))
    resync_levels("snac_encode", lambda f: (lambda o: (o[1], o[2]))(eng.quantizer(torch.from_numpy(fx["z"]), return_margins=True, force=f)), want, wm, thr=2e-3, what=f"reference run {kind}")
    if all(torch.equal(a.cpu(), b) for a, b in zip(codes, want)):
        assert rel_peak(z_q, fx["z_q"]) < 1e-5
    got = eng.encode(audio)
    assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(got, want))
    hat, codes2 = eng(audio)
    torch.cuda.synchronize()
    assert hat.shape[0] == 2 and hat.shape[2] == 1 and torch.isfinite(hat).all() and all(torch.equal(a, b) for a, b in zip(got, codes2))


@pytest.mark.parametrize("depthwise", [True, False])
def test_snac_encode_stages_and_codes_vs_oracle(depthwise):
    """24 kHz-model strides (2 / 4 / 8 / 8) and levels (4 / 2 / 1) at widths that take the wave-specialised kernel, fp16-exact folded weights: every
    encoder stage, then the free-running codes under the margin rule."""
    from mlx_audio_amd.codec.models.snac import SNAC, make_snac_encoder_weights, make_snac_weights
    from oracle.snac_ref import SNACEncoderRef

    c = dict(sampling_rate=24000, encoder_dim=16, encoder_rates=[2, 4, 8, 8], decoder_dim=256, decoder_rates=[8, 8, 4, 2], attn_window_size=None, codebook_size=1024,
             codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=depthwise)
    latent = 256
    w = make_snac_weights(latent, 256, c["decoder_rates"], c["vq_strides"], 1024, 8, True, depthwise, seed=5)
    w.update(make_snac_encoder_weights(16, c["encoder_rates"], latent, c["vq_strides"], 8, depthwise, seed=5))
    fp16_exact_snac(w)
    eng = SNAC(**c, weights=w, device=DEV)
    ref = SNACEncoderRef(w, c["encoder_rates"], c["vq_strides"], depthwise=depthwise)
    audio = make_audio(2, 512 * 4 * 11 + 77, 24000, seed=8)
    padded = ref.preprocess(audio)
    assert torch.equal(eng.preprocess(audio), padded) and padded.shape[-1] == 512 * 4 * 12
    zr, est = ref.encoder(padded, return_stages=True)
    z, gst = eng.encoder(padded, return_stages=True)
    torch.cuda.synchronize()
    errs = {k: rel_peak(gst[k], est[k]) for k in est}
    print(f"snac encoder depthwise={depthwise}: stage rel err { {k: f'{v:.1e}' for k, v in errs.items()} }")
    assert max(errs.values()) < 2e-4, errs
    _, wc, wm = ref.quantize(zr, return_margins=True)
    got, gm = eng.encode(audio, return_margins=True)
    torch.cuda.synchronize()
    resync_levels("snac_encode", lambda f: eng.encode(audio, return_margins=True, force=f), wc, wm, thr=1e-5, what=f"depthwise={depthwise}")   # measured gap difference 1.3e-6
    agree = float(torch.cat([(a.cpu() == b).float().flatten() for a, b in zip(got, wc)]).mean())
    print(f"snac encode depthwise={depthwise}: {100 * agree:.1f} % of all codes equal the oracle's (free-running)")
    assert agree > 0.85, agree
    y = eng.decode(got)
    torch.cuda.synchronize()
    assert y.shape[0] == 2 and y.shape[2] == 1 and torch.isfinite(y).all()


# ------------------------------------------------------------------------------------------------------------------ EnCodec
@pytest.mark.parametrize("tag", ["mono", "stereo"])
def test_encodec_encode_against_the_reference_run(tag):
    """The reference's own ``Encodec.encode`` outputs (tests/golden/ref_encodec_encode_*.npz): the encoder's embeddings, then every code of both
    bandwidths under the margin rule -- mono causal one-chunk, and stereo non-causal with loudness normalisation and overlapping chunks."""
    from mlx_audio_amd.codec.models.encodec import Encodec
    from test_codec_encode_cpu import encodec_model_weights
    from oracle.encodec_ref import EncodecRef

    fx = np.load(os.path.join(GOLD, f"ref_encodec_encode_{tag}.npz"))
    c, w = encodec_model_weights(fx)
    eng, ref = Encodec(c, weights=w, device=DEV), EncodecRef(w, c)
    x, m = torch.from_numpy(fx["inputs"]), torch.from_numpy(fx["masks"])
    chunk = eng.chunk_length or x.shape[1]
    emb = eng._encoder(x[:, :chunk])
    torch.cuda.synchronize()
    e = rel_peak(emb, fx["embeddings_chunk0_unnormalised"])
    print(f"encodec ({tag}) encoder vs the reference run: {e:.2e} of the peak")
    assert tuple(emb.shape) == fx["embeddings_chunk0_unnormalised"].shape and e < 2e-3, e
    bw = c["target_bandwidths"][-1]
    want = torch.from_numpy(fx[f"codes_bw{bw}"]).long()                  # [chunks, B, nq, T]
    codes, scales = eng.encode(x, m, bandwidth=bw)
    torch.cuda.synchronize()
    assert tuple(codes.shape) == tuple(want.shape) and codes.dtype == torch.int64
    if c["normalize"]:
        assert rel_peak(torch.stack(scales), fx[f"scales_bw{bw}"]) < 1e-5
    # the oracle's margins per chunk (its codes ARE the fixture's: tests/test_codec_encode_cpu.py), the device's from a second search
    step = chunk - (eng.chunk_stride or chunk)
    for ci, off in enumerate(range(0, x.shape[1] - step, eng.chunk_stride or chunk)):
        xc, mc = x[:, off:off + chunk], m[:, off:off + chunk]
        if c["normalize"]:
            xc = xc * mc[..., None].float()
            mono = xc.sum(dim=2, keepdim=True) / xc.shape[2]
            xc = xc / (torch.sqrt((mono ** 2).mean(dim=1, keepdim=True)) + 1e-8)
        er = ref.encoder(xc)
        wc, wm = ref.quantizer_encode(er, bw, return_margins=True)
        assert torch.equal(wc, want[ci])
        gc, gm = eng.quantizer.encode(eng._encoder(xc), bw, return_margins=True)
        assert torch.equal(gc.cpu(), codes[ci].cpu())
        thr = 2e-3 * float(er.abs().max()) * 3.0
        eemb = eng._encoder(xc)
        resync_frames("encodec_encode", lambda f: eng.quantizer.encode(eemb, bw, return_margins=True, force=f), want[ci], wm, thr=thr, what=f"reference run {tag} chunk {ci}")
        # the per-layer forced path with nothing forced IS the one-launch path
        nf, _ = eng.quantizer.encode(eemb, bw, return_margins=True, force=(torch.zeros_like(want[ci], dtype=torch.bool), want[ci]))
        assert torch.equal(nf.cpu(), gc.cpu())
    lo = c["target_bandwidths"][0]
    codes_lo, _ = eng.encode(x, m, bandwidth=lo)
    assert tuple(codes_lo.shape) == fx[f"codes_bw{lo}"].shape
    audio = eng.decode(codes, scales, m)
    torch.cuda.synchronize()
    assert tuple(audio.shape) == fx["decoded"].shape and torch.isfinite(audio).all()


def test_encodec_24khz_encode_stages_and_codes_vs_oracle():
    """The published 24 kHz configuration (32 filters, ratios 8 / 5 / 4 / 2, two LSTM layers of 512, 1024 x 128 codebooks, 24 kbps = 32 quantizers), weights
    on the fp16 grid (the image is then exact): every encoder stage against the float32 oracle, then the free-running codes under the margin rule, and the
    encode -> decode round trip."""
    from mlx_audio_amd.codec.models.encodec import Encodec, make_encodec_encoder_weights, make_encodec_weights
    from oracle.encodec_ref import EncodecRef

    c = dict(upsampling_ratios=[8, 5, 4, 2], target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0])
    w = make_encodec_weights(c, seed=9)
    w.update(make_encodec_encoder_weights(c, seed=9))
    w = {k: (v.half().float() if v.is_floating_point() and "codebook" not in k else v) for k, v in w.items()}
    eng, ref = Encodec(c, weights=w, device=DEV), EncodecRef(w, c)
    x = make_audio(1, 24000 + 123, 24000, seed=4).transpose(1, 2).contiguous()      # [1, samples, 1]
    er, est = ref.encoder(x, return_stages=True)
    eg, gst = eng._encoder(x, return_stages=True)
    torch.cuda.synchronize()
    errs = {k: rel_peak(gst[k], est[k]) for k in est}
    print(f"encodec 24 kHz encoder: stage rel err { {k: f'{v:.1e}' for k, v in errs.items()} }")
    assert tuple(eg.shape) == tuple(er.shape) == (1, 76, 128) and max(errs.values()) < 3e-4, errs
    wc, wm = ref.quantizer_encode(er, 24.0, return_margins=True)
    gc, gm = eng.quantizer.encode(eg, 24.0, return_margins=True)
    torch.cuda.synchronize()
    assert tuple(gc.shape) == (1, 32, 76)
    resync_frames("encodec_encode", lambda f: eng.quantizer.encode(eg, 24.0, return_margins=True, force=f), wc, wm, thr=2.5e-5 * float(er.abs().max()) * 3.0, what="24 kHz, 32 layers")   # measured gap difference 5.7e-5 (threshold 3e-4: 5x)
    agree = float((gc.cpu() == wc).float().mean())
    print(f"encodec 24 kHz: {100 * agree:.1f} % of all codes equal the oracle's (free-running, 32 layers deep)")
    assert agree > 0.5, agree
    codes, scales = eng.encode(x, None, bandwidth=24.0)
    assert tuple(codes.shape) == (1, 1, 32, 76) and scales == [None] and torch.equal(codes[0].cpu(), gc.cpu())
    audio = eng.decode(codes, scales)
    torch.cuda.synchronize()
    assert tuple(audio.shape) == (1, 76 * 320, 1) and torch.isfinite(audio).all()
    for bw, nq in ((1.5, 2), (6.0, 8)):
        assert tuple(eng.encode(x, None, bandwidth=bw)[0].shape) == (1, 1, nq, 76)


def test_vocos_with_encodec_features():
    """``Vocos`` over ``EncodecFeatures`` (codec/models/vocos/vocos.py:54-116, 350-375): audio -> EnCodec codes -> summed codebook rows -> backbone -> iSTFT.
    ``__call__`` equals ``decode(features)`` and ``decode_from_codes(codes)``; the codes are the EnCodec engine's own."""
    from mlx_audio_amd.codec.models.encodec import Encodec
    from mlx_audio_amd.codec.models.vocos import EncodecFeatures, Vocos
    from test_codec_encode_cpu import encodec_model_weights

    fx = np.load(os.path.join(GOLD, "ref_encodec_encode_mono.npz"))
    c, w = encodec_model_weights(fx)
    enc = Encodec(c, weights=w, device=DEV)
    cfg = {"feature_extractor": {"class_path": "vocos.feature_extractors.EncodecFeatures", "init_args": {"encodec_model": "encodec_24khz", "bandwidths": c["target_bandwidths"]}},
           "backbone": {"class_path": "vocos.models.VocosBackbone",
                        "init_args": {"input_channels": c["codebook_dim"], "dim": 64, "intermediate_dim": 192, "num_layers": 2, "adanorm_num_embeddings": 2}},
           "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 64, "n_fft": 64, "hop_length": 16, "padding": "same"}}}
    v = Vocos.from_hparams(cfg, device=DEV, seed=3, encodec=enc)
    assert isinstance(v.feature_extractor, EncodecFeatures) and v.feature_extractor.num_q == 6   # floor(60 kbps / 9 kbps per layer); the model holds 4
    audio = torch.from_numpy(fx["raw"][:, 0])
    bw = torch.tensor([[1.0, 1.0]])                                  # index 1 for the extractor, the conditioning vector of the AdaLayerNorms
    codes = v.get_encodec_codes(audio, bandwidth_id=bw)
    want, _ = enc.encode(*v.feature_extractor.preprocessor(audio), bandwidth=c["target_bandwidths"][1])
    assert tuple(codes.shape) == (4, 1, want.shape[-1]) and torch.equal(codes[:, 0], want[0, 0])
    feats = v.feature_extractor.get_features_from_codes(codes)
    rows = sum(w[f"quantizer.layers.{i}.codebook.embed"][codes[i, 0].cpu()] for i in range(4))
    assert rel_peak(feats[0], rows) < 1e-6
    y = v(audio, bandwidth_id=bw)
    y2 = v.decode(feats, bandwidth_id=bw)
    y3 = v.decode_from_codes(codes, bandwidth_id=bw)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and y.numel() >= (want.shape[-1] - 1) * 16 and torch.equal(y, y2) and torch.equal(y, y3)
    none = Vocos.from_hparams(cfg, device=DEV, seed=3)             # hub name, no model: decode works, __call__ is loud
    assert none.feature_extractor is None and torch.equal(none.decode(feats, bandwidth_id=bw), y)
    with pytest.raises(FileNotFoundError):
        none(audio, bandwidth_id=bw)


# ------------------------------------------------------------------------------------------------------------------ the search kernel
@pytest.mark.parametrize("D,bins,layers", [(128, 1024, 8), (8, 1024, 1), (256, 2048, 5), (32, 64, 4)])
def test_rvq_encode_frames_per_workgroup_forms_are_bit_identical(D, bins, layers):
    """``mi355_rvq_encode`` takes 8 / 4 / 2 frames per workgroup from 4096 / 2048 / 1024 frames on (round 5: the frames of a workgroup share each read of the
    layer's table) and one below: the same frames searched in one large call and in calls of < 1024 frames give the same codes AND the same margins, bit for
    bit (a thread walks d in the same order with the same fmaf in every form); and both equal the float32 restatement outside knife edges."""
    from mlx_audio_amd import ops

    g = torch.Generator().manual_seed(D + bins + layers)
    tables = (torch.randn(layers, bins, D, generator=g) / torch.arange(1, layers + 1).sqrt()[:, None, None]).contiguous()
    tt, c2 = tables.transpose(1, 2).contiguous(), ((tables * tables).sum(-1) / 2).contiguous()
    td, ttd, c2d = tables.to(DEV), tt.to(DEV), c2.to(DEV)
    for n in (5003, 2500, 1500):
        x = torch.randn(n, D, generator=g) * 1.5
        xd = x.to(DEV)
        big, bm = ops.rvq_encode(xd, td, ttd, c2d, margins=True)
        parts = [ops.rvq_encode(xd[i:i + 700], td, ttd, c2d, margins=True) for i in range(0, n, 700)]
        torch.cuda.synchronize()
        small, sm = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        assert torch.equal(big, small) and torch.equal(bm, sm), (n, int((big != small).sum()))
    # against the float32 restatement (first minimum of |e|^2 / 2 - x . e, residual -= e), away from knife edges
    r = x.clone()
    for l in range(layers):
        s = c2[l][None, :] - r @ tt[l]
        idx = s.argmin(1)
        clear = bm[:, l].cpu() > 1e-4 * float(s.abs().max())
        assert torch.equal(big[:, l].cpu()[clear].long(), idx[clear]), l
        r = r - tables[l][big[:, l].cpu().long()]


# ------------------------------------------------------------------------------------------------------------------ edge cases
def test_codec_encode_edge_cases():
    """Smallest and ragged shapes of the encode sides: one utterance, exactly one hop, a length that is not a whole number of hops (DAC encodes the raw
    length, dac.py:184-192: the strided convs floor), SNAC's minimal padded length, EnCodec in a BATCH (the reference's Metal LSTM kernel is only
    consistent for one sequence; the oracle runs the per-sequence LSTM it means) -- a batch equals its items wherever no knife edge is involved."""
    from mlx_audio_amd.codec.models.encodec import Encodec
    from mlx_audio_amd.codec.models.snac import SNAC
    from test_codec_encode_cpu import encodec_model_weights, snac_model_weights
    from oracle.encodec_ref import EncodecRef
    from oracle.snac_ref import SNACEncoderRef

    # DAC
    c = dict(encoder_dim=32, encoder_rates=[2, 4, 5, 8], latent_dim=64, decoder_dim=64, decoder_rates=[8, 5, 4, 2], n_codebooks=3, codebook_size=256, codebook_dim=8, sample_rate=16000)
    eng, ref, _ = dac_pair(c, 23, True)
    for S in (320, 320 * 3 + 1, 320 * 2 + 319):
        a = make_audio(1, S, 16000, seed=S)
        want = ref.quantize(ref.encoder(a), return_margins=True)
        got = eng.encode(a, return_margins=True)
        torch.cuda.synchronize()
        assert tuple(got[1].shape) == tuple(want[1].shape) and (S % 320 or got[1].shape[2] == S // 320), (S, tuple(got[1].shape), tuple(want[1].shape))   # 959 samples: 3 frames (each strided conv floors)
        resync_frames("dac_encode", lambda f: (lambda o: (o[1], o[5]))(eng.encode(a, return_margins=True, force=f)), want[1], want[5], thr=1e-5, what=f"{S} samples")   # exact images: measured gap difference < 6e-7
        assert rel_peak(got[2], want[2]) < 1e-3 or not torch.equal(got[1].cpu(), want[1])
    # SNAC: the shortest input (one sample) pads to hop * lcm(vq_strides) samples = lcm finest frames
    fx = np.load(os.path.join(GOLD, "ref_snac_encode_dw.npz"))
    sc, sw = snac_model_weights(fx)
    sn, sref = SNAC(**sc, weights=sw, device=DEV), SNACEncoderRef(sw, sc["encoder_rates"], sc["vq_strides"], depthwise=True)
    one = torch.full((1, 1, 1), 0.25)
    got, want = sn.encode(one), sref.encode(one)
    assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want] == [(1, 1), (1, 2), (1, 4)]
    # EnCodec: three utterances at once vs one by one
    fe = np.load(os.path.join(GOLD, "ref_encodec_encode_mono.npz"))
    ec, ew = encodec_model_weights(fe)
    en, eref = Encodec(ec, weights=ew, device=DEV), EncodecRef(ew, ec)
    x = make_audio(3, 16 * 50 + 7, 24000, seed=2).transpose(1, 2).contiguous()
    bw = ec["target_bandwidths"][-1]
    both, _ = en.encode(x, None, bandwidth=bw)
    torch.cuda.synchronize()
    assert tuple(both.shape) == (1, 3, 4, 51)
    for b in range(3):
        alone, _ = en.encode(x[b:b + 1], None, bandwidth=bw)
        want, _ = eref.encode(x[b:b + 1], None, bandwidth=bw)
        eq_batch = float((both[0, b].cpu() == alone[0, 0].cpu()).float().mean())
        eq_ref = float((alone[0, 0].cpu() == want[0, 0]).float().mean())
        # a float32 checkpoint behind an fp16 image moves the embeddings by ~5e-4: a flipped code changes every later layer of its frame
        assert eq_batch > 0.95 and eq_ref > 0.85, (b, eq_batch, eq_ref)


def test_snac_encode_with_local_mha_vs_oracle():
    """The 32 / 44 kHz SNAC family: ``LocalMHA`` between the last ``EncoderBlock`` and the final conv of the ENCODER (layers.py:148-149).  Like the decoder's
    (tests/test_snac_gpu.py::test_local_mha_variant_vs_oracle) it is held to the oracle's restatement of what the module means -- the reference's own
    transcription raises on channels-last data (tests/golden/ref_snac_local_mha_probe.json): PARITY UNPINNED for this one stage."""
    from mlx_audio_amd.codec.models.snac import SNAC, make_snac_encoder_weights, make_snac_weights
    from oracle.snac_ref import SNACEncoderRef

    c = dict(sampling_rate=32000, encoder_dim=16, encoder_rates=[2, 4, 8], decoder_dim=128, decoder_rates=[8, 4, 2], attn_window_size=4, codebook_size=256,
             codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True)
    latent = 128
    w = make_snac_weights(latent, 128, c["decoder_rates"], c["vq_strides"], 256, 8, True, True, seed=12, attn=True)
    w.update(make_snac_encoder_weights(16, c["encoder_rates"], latent, c["vq_strides"], 8, True, seed=12, attn=True))
    fp16_exact_snac(w)
    for k in [k for k in w if k.endswith(("to_qkv.weight", "to_out.weight"))]:
        w[k] = w[k].half().float()
    eng = SNAC(**c, weights=w, device=DEV)
    ref = SNACEncoderRef(w, c["encoder_rates"], c["vq_strides"], depthwise=True, attn_window_size=4)
    audio = make_audio(2, 64 * 4 * 7 + 19, 32000, seed=6)
    padded = ref.preprocess(audio)
    assert padded.shape[-1] == 64 * 4 * 8     # hop 64 x lcm(strides 4 / 2 / 1, window 4) = 256
    zr, est = ref.encoder(padded, return_stages=True)
    z, gst = eng.encoder(padded, return_stages=True)
    torch.cuda.synchronize()
    errs = {k: rel_peak(gst[k], est[k]) for k in est}
    print(f"snac encoder with LocalMHA: stage rel err { {k: f'{v:.1e}' for k, v in errs.items()} }")
    assert "attn" in errs and max(errs.values()) < 3e-4, errs
    _, wc, wm = ref.quantize(zr, return_margins=True)
    resync_levels("snac_encode", lambda f: eng.encode(audio, return_margins=True, force=f), wc, wm, thr=1e-5, what="LocalMHA")   # measured gap difference 9.2e-7


# ------------------------------------------------------------------------------------------------------------------ the reference's own test files
@pytest.mark.parametrize("sr,n_samples,enc_rates,dec_rates,nq,frames,out_len", [
    (16_000, 80_000, [2, 4, 5, 8], [8, 5, 4, 2], 12, 250, 80_043),
    (24_000, 120_000, [2, 4, 5, 8], [8, 5, 4, 2], 32, 375, 120_043),
    (44_100, 220_000, [2, 4, 8, 8], [8, 8, 4, 2], 9, 430, 220_235),
])
def test_reference_test_descript_as_written(sr, n_samples, enc_rates, dec_rates, nq, frames, out_len):
    """codec/tests/test_descript.py:13-109, statement for statement (``mx.zeros`` -> ``torch.zeros``): a FRESHLY CONSTRUCTED model -- published widths, both
    halves -- preprocesses, encodes, decodes with the shapes the reference pins."""
    from mlx_audio_amd.codec.models.descript import DAC

    audio = torch.zeros((1, 1, n_samples))
    model = DAC(encoder_dim=64, encoder_rates=enc_rates, decoder_dim=1536, decoder_rates=dec_rates, n_codebooks=nq, codebook_size=1024, codebook_dim=8, sample_rate=sr)
    x = model.preprocess(audio, sr)
    z, codes, latents, _, _ = model.encode(x)
    assert tuple(z.shape) == (1, 1024, frames) and tuple(codes.shape) == (1, nq, frames) and tuple(latents.shape) == (1, 8 * nq, frames)
    y = model.decode(z).squeeze(-1)
    assert tuple(y.shape) == (1, out_len) and torch.isfinite(y).all()


def test_reference_test_snac_as_written():
    """codec/tests/test_snac.py:8-38."""
    from mlx_audio_amd.codec.models.snac import SNAC

    config = {"sampling_rate": 24000, "encoder_dim": 48, "encoder_rates": [2, 4, 8, 8], "decoder_dim": 1024, "decoder_rates": [8, 8, 4, 2], "attn_window_size": None,
              "codebook_size": 4096, "codebook_dim": 8, "vq_strides": [4, 2, 1], "noise": True, "depthwise": True}
    audio = torch.zeros((1, 1, 120_000))
    model = SNAC(**config)
    codes = model.encode(audio)
    assert len(codes) == 3 and tuple(codes[0].shape) == (1, 59) and tuple(codes[1].shape) == (1, 118) and tuple(codes[2].shape) == (1, 236)
    reconstructed = model.decode(codes).squeeze(-1)
    assert tuple(reconstructed.shape) == (1, 120_907) and torch.isfinite(reconstructed).all()


def test_reference_test_encodec_as_written():
    """codec/tests/test_encodec.py:7-57."""
    from mlx_audio_amd.codec.models.encodec import Encodec, EncodecConfig

    config = EncodecConfig(audio_channels=1, chunk_length_s=None, codebook_dim=128, codebook_size=1024, compress=2, dilation_growth_rate=2, hidden_size=128,
                           kernel_size=7, last_kernel_size=7, model_type="encodec", norm_type="weight_norm", normalize=False, num_filters=32, num_lstm_layers=2,
                           num_residual_layers=1, overlap=None, pad_mode="reflect", residual_kernel_size=3, sampling_rate=24000,
                           target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0], trim_right_ratio=1.0, upsampling_ratios=[8, 5, 4, 2], use_causal_conv=True)
    model = Encodec(config)
    audio = torch.zeros((1, 120_000, 1))
    codes, scales = model.encode(audio)                      # default bandwidth
    assert tuple(codes.shape) == (1, 1, 2, 375)
    audio_out = model.decode(codes, scales)
    assert tuple(audio_out.shape) == (1, 120_000, 1)
    codes, scales = model.encode(audio, bandwidth=6)         # 6 kbps
    assert tuple(codes.shape) == (1, 1, 8, 375)
    audio_out = model.decode(codes, scales)
    assert tuple(audio_out.shape) == (1, 120_000, 1) and torch.isfinite(audio_out).all()


def test_reference_test_vocos_encodec_half_as_written():
    """codec/tests/test_vocos.py:33-56, 77-92, as written except that the EnCodec model comes in by argument (the reference's ``EncodecFeatures`` downloads
    it from the hub by name): 120 000 samples -> 24 kHz EnCodec codes at ``bandwidths[3]`` -> summed codebook rows -> AdaLayerNorm backbone -> iSTFT head
    (n_fft 1280, hop 320, 'same' padding): 119 680 samples both through ``__call__`` and through ``get_encodec_codes`` + ``decode_from_codes``."""
    from mlx_audio_amd.codec.models.encodec import Encodec, EncodecConfig
    from mlx_audio_amd.codec.models.vocos import Vocos

    config_encodec = {
        "feature_extractor": {"class_path": "vocos.feature_extractors.EncodecFeatures",
                              "init_args": {"encodec_model": "encodec_24khz", "bandwidths": [1.5, 3.0, 6.0, 12.0, 24.0]}},
        "backbone": {"class_path": "vocos.models.VocosBackbone",
                     "init_args": {"input_channels": 128, "dim": 384, "intermediate_dim": 1152, "num_layers": 8, "adanorm_num_embeddings": 4}},
        "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 384, "n_fft": 1280, "hop_length": 320, "padding": "same"}},
    }
    encodec_24khz = Encodec(EncodecConfig(upsampling_ratios=[8, 5, 4, 2], target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0]))
    audio = torch.zeros((120_000))
    model = Vocos.from_hparams(config_encodec, encodec=encodec_24khz)
    bandwidth_id = [3, 3, 3, 3]
    reconstructed_audio = model(audio, bandwidth_id=torch.tensor(bandwidth_id)[None, ...])
    assert tuple(reconstructed_audio.shape) == (119680,)
    codes = model.get_encodec_codes(audio, bandwidth_id=bandwidth_id)
    assert tuple(codes.shape) == (16, 1, 375)                     # 12 kbps = 16 codebooks
    decoded = model.decode_from_codes(codes, bandwidth_id=torch.tensor(bandwidth_id)[None, ...])
    assert tuple(decoded.shape) == (119680,) and torch.equal(decoded, reconstructed_audio) and torch.isfinite(decoded).all()


def test_dac_compress_decompress_against_the_reference_run():
    """``DAC.compress`` / ``decompress`` (codec/models/descript/base.py:123-231) on the device against the reference's own run (scripted audio reader,
    tests/golden/ref_dac_compress.npz): chunking, normalisation, the codes of every chunk under the margin rule, the reconstruction where the codes agree."""
    from mlx_audio_amd.codec.models.descript import DAC
    from test_codec_encode_cpu import dac_model_weights
    from oracle.dac_ref import DACEncoderRef

    fx = np.load(os.path.join(GOLD, "ref_dac_compress.npz"))
    c, w = dac_model_weights(fx)
    eng, ref = DAC(**c, weights=w, device=DEV), DACEncoderRef(w, c["encoder_rates"], c["n_codebooks"])
    for tag in ("long", "short"):
        f = eng.compress((fx[f"{tag}_signal"], c["sample_rate"]), win_duration=float(fx[f"{tag}_win"]))
        want = torch.from_numpy(fx[f"{tag}_codes"]).long()
        assert tuple(f.codes.shape) == tuple(want.shape) and f.chunk_length == int(fx[f"{tag}_chunk_length"]) and f.padding == bool(fx[f"{tag}_padding"])
        assert abs(f.input_db - float(fx[f"{tag}_input_db"])) < 1e-4 and eng.padding is True
        same = (f.codes.cpu() == want).all(dim=1)[0]                  # frames whose whole chain agrees (a float32 checkpoint behind an fp16 image)
        assert float(same.float().mean()) > 0.7, float(same.float().mean())
        rec = eng.decompress(f)
        torch.cuda.synchronize()
        assert tuple(rec.shape) == fx[f"{tag}_recons"].shape and torch.isfinite(rec).all()
        if bool(same.all()):
            assert rel_peak(rec, fx[f"{tag}_recons"]) < 2e-3
