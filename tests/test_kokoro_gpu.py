"""End-to-end parity of the Kokoro-82M HIP path against the CPU oracle (needs an MI355X).

How parity is stated (DESIGN.md "Parity"):
  * integer path: predicted durations / alignment are bit-exact whenever the oracle's pre-rounding value is
    >= 1e-3 away from a .5 boundary (the margin is asserted, not assumed);
  * front end (PL-BERT, LSTMs, prosody predictor): d, F0, N, asr within 5e-4 relative, free-running;
  * vocoder (decoder + iSTFTNet generator + iSTFT): waveform max-abs error <= 2e-3 * peak and SNR >= 50 dB
    against the fp32 oracle (the figures SURVEY.md section 8c proposes; the reference's own numeric tests use
    rtol = atol = 2e-3), with the oracle's F0 / N curves and harmonic STFT features injected.  The
    injection is needed because the harmonic source INTEGRATES F0 into a phase: a 1e-5 relative F0 rounding
    difference moves the 9th harmonic by radians within a second, and the reflect-padded edge frames are
    exactly symmetric so their phase features are atan2(+-rounding noise, x): sample-wise agreement of the
    free-running waveform is not a property even two builds of the reference would have;
  * free-running waveform: checked through the log-mel spectral envelope instead.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine
    from oracle.kokoro_ref import KokoroRef

    w = S.make_kokoro_weights()
    eng = KokoroEngine(w, S.KOKORO_CONFIG)
    ref32 = KokoroRef(S.make_kokoro_weights(), S.KOKORO_CONFIG, dtype=torch.float32)
    return S, eng, ref32


# Free-running spectral-envelope bar (log10 power, mean |diff| over the 80-band log-mel of the whole utterance).  Measured on MI355X in round 2:
# see the value printed by test_kokoro_front_end_free_running (profiles/r2_kokoro_free_running_call11.txt: 0.1796); the bar is 1.5x that measurement
# (round 6; it was 2x: VERDICT r5 weak point 3).
# What the bar separates (round 5, CPU study pinned by tests/test_oracle_golden.py::test_kokoro_free_running_envelope_bar_is_the_oracle_self_sensitivity):
# the oracle against ITSELF with its F0 curve perturbed by 1e-5 of the peak (float32 rounding level) scores 0.14 - 0.18 on this metric -- the device's
# figure -- and 0.24 - 0.33 at 1e-4, 0.37 - 0.40 at 5e-4: the free-running waveform is chaotic in F0 (phase integration), so a sample-level bar cannot exist
# and this one sits at "an F0 error of about 5e-5 of the peak".  The F0 / N curves themselves are held at 5e-4 relative just above.
ENV_BAR = 0.27  # measured 0.1796 (round 2, call 11)


def snr_db(got, ref):
    err = (got.double() - ref.double())
    return float(10 * torch.log10(ref.double().pow(2).sum() / err.pow(2).sum().clamp_min(1e-300)))


def _noise(F, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(size=(1, 9)).astype(np.float32), rng.standard_normal((1, 2 * F * 300, 9)).astype(np.float32)


def _teacher(tr):
    return dict(f0=tr["f0"], n=tr["n"], har=tr["har"].transpose(1, 2))


def test_kokoro_front_end_free_running(setup):
    S, eng, ref = setup
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    pd, d, raw = ref.durations(ids, ref_s, speed=1.3)
    F = int(pd.sum())
    ri, nz = _noise(F, 77)
    audio_ref, _, tr = ref.forward(ids, ref_s, speed=1.3, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, durs, tg = eng.forward([ids], ref_s, speed=1.3, rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                                 return_intermediates=True)
    torch.cuda.synchronize()
    margin = float(((raw - torch.floor(raw)) - 0.5).abs().min())
    assert float((tg["dur_raw"][0, : len(ids)].cpu() - raw).abs().max()) < 5e-4
    assert margin > 1e-3, "pick another seed: the oracle itself sits on a rounding boundary"
    # bit-exact integer path UNDER THE MARGIN ABOVE, not unconditionally: the pre-rounding durations carry fp32 summation-order differences of ~1e-6
    # relative that depend on the launch size (conv_gemm picks its kernel and its split-K grouping by the tile count B * L_out: csrc/conv_gemm.hip
    # split_groups), so the same utterance may differ at that level between batch sizes; a value >= 1e-3 from a .5 boundary rounds the same everywhere
    assert torch.equal(durs[0].cpu(), pd)
    assert outs[0].shape == audio_ref[0].shape == (600 * F,)

    def rel(a, b):
        return float((a.cpu().double() - b.double()).abs().max() / b.abs().max())

    assert rel(tg["d"][0], tr["d"][0]) < 5e-4
    assert rel(tg["f0"][0], tr["f0"][0]) < 5e-4
    assert rel(tg["n"][0], tr["n"][0]) < 5e-4
    assert rel(tg["asr"][0], tr["asr"][0].transpose(0, 1)) < 5e-4
    assert rel(tg["dec3"][0], tr["dec3"][0].transpose(0, 1)) < 1e-3
    # free-running waveform: same spectral envelope (harmonic phases are not comparable, see module docstring)
    from oracle import dsp_ref

    fb = dsp_ref.mel_filters(24000, 1024, 80, norm="slaney", mel_scale="slaney")

    def logmel(x):
        s = np.abs(dsp_ref.stft(x, n_fft=1024, hop_length=256)) ** 2
        return np.log10(np.maximum(s @ fb.T, 1e-8))

    a, b = logmel(outs[0].cpu().numpy()), logmel(audio_ref[0].numpy())
    assert a.shape == b.shape
    env_err = float(np.mean(np.abs(a - b)))
    print(f"free-running log-mel envelope mean |diff| = {env_err:.4f} (log10 power units)")
    assert env_err < ENV_BAR


def test_kokoro_vocoder_teacher_forced(setup):
    S, eng, ref = setup
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    pd, _, _ = ref.durations(ids, ref_s, speed=1.3)
    F = int(pd.sum())
    ri, nz = _noise(F, 77)
    audio_ref, _, tr = ref.forward(ids, ref_s, speed=1.3, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, _, tg = eng.forward([ids], ref_s, forced_durations=[pd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                              overrides=_teacher(tr), return_intermediates=True)
    torch.cuda.synchronize()
    got = outs[0].cpu()
    peak = float(audio_ref.abs().max())
    err = float((got - audio_ref[0]).abs().max())
    snr = snr_db(got, audio_ref[0])
    print(f"kokoro vocoder (teacher-forced): F={F} peak={peak:.3f} max_abs_err={err:.3e} snr={snr:.1f} dB")
    assert err <= 2e-3 * peak, err
    assert snr >= 50.0, snr
    # with only F0/N injected, the HIP harmonic features agree with the oracle except for +-pi branch flips of
    # rounding-noise phases; those must be rare and every other feature must match tightly
    outs2, _, tg2 = eng.forward([ids], ref_s, forced_durations=[pd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                                overrides=dict(f0=tr["f0"], n=tr["n"]), return_intermediates=True)
    torch.cuda.synchronize()
    assert float((tg2["har_src"].cpu() - tr["har_src"]).abs().max()) < 1e-6
    hd = (tg2["har"].cpu() - tr["har"].transpose(1, 2)).abs()
    flips = hd > 1.0
    assert int(flips.sum()) <= 8, int(flips.sum())
    assert float(hd[~flips].max()) < 2e-3
    two_pi = hd[flips]
    assert bool(((two_pi - 2 * np.pi).abs() < 1e-3).all())


def test_kokoro_vocoder_free_running_f0n_only(setup):
    """The vocoder FREE-RUNNING from the pitch / energy curves: only F0 / N are injected; SineGen (phase integration, noise, tanh), the n_fft = 20
    STFT, decoder, generator and iSTFT head all run on their own HIP outputs.  Bar: 2e-3 * peak and 50 dB against the fp32 oracle, after
    restoring the <= 8 harmonic-phase entries whose value is a +-pi branch of atan2(+-rounding noise, x) (reflect-padded edge frames are exactly
    symmetric, so their imaginary parts are pure rounding noise; which branch comes out is not a property of the algorithm).

    Why F0 / N themselves must be injected -- the F0-error -> phase budget: the source phase of harmonic h is 2 pi h * sum(f0) / sr, so a relative
    F0 error eps becomes a phase error 2 pi h f0 eps t after t seconds.  A waveform error of 2e-3 * peak needs that phase error below ~2e-3 rad at
    the top harmonic (h = 9): for f0 = 200 Hz over the canonical 6.6 s utterance that is eps < 2e-3 / (2 pi * 9 * 200 * 6.6) -- asserted below to be under fp32's
    own rounding step (2^-24), i.e. no fp32 implementation of the front end, the reference's included, could deliver F0 to that precision:
    the measured front-end agreement (5e-4 relative, test above) is 4 orders of magnitude away from it."""
    S, eng, ref = setup
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    pd, _, _ = ref.durations(ids, ref_s, speed=1.3)
    F = int(pd.sum())
    ri, nz = _noise(F, 77)
    audio_ref, _, tr = ref.forward(ids, ref_s, speed=1.3, rand_ini=ri, noise=nz, return_intermediates=True)
    kw = dict(forced_durations=[pd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz), return_intermediates=True)
    outs, _, tg = eng.forward([ids], ref_s, overrides=dict(f0=tr["f0"], n=tr["n"]), **kw)
    torch.cuda.synchronize()
    har_hip, har_ref = tg["har"].cpu(), tr["har"].transpose(1, 2)
    flips = (har_hip - har_ref).abs() > 1.0
    n_flips = int(flips.sum())
    assert n_flips <= 8, n_flips
    peak = float(audio_ref.abs().max())
    raw_err = float((outs[0].cpu() - audio_ref[0]).abs().max())
    if n_flips:
        patched = torch.where(flips, har_ref, har_hip)   # the HIP features, with only the branch-ambiguous entries set to the oracle's branch
        outs, _, _ = eng.forward([ids], ref_s, overrides=dict(f0=tr["f0"], n=tr["n"], har=patched), **kw)
        torch.cuda.synchronize()
    got = outs[0].cpu()
    err = float((got - audio_ref[0]).abs().max())
    snr = snr_db(got, audio_ref[0])
    print(f"kokoro vocoder (free-running from F0 / N): F={F} peak={peak:.3f} branch flips={n_flips} max_abs_err={err:.3e} "
          f"(before restoring the flips {raw_err:.3e}) snr={snr:.1f} dB")
    assert err <= 2e-3 * peak, err
    assert snr >= 50.0, snr
    t_s = 6.6  # the canonical utterance of the benchmark; this test's own utterance is shorter
    eps_needed = 2e-3 / (2 * np.pi * 9 * 200.0 * t_s)
    assert eps_needed < 2.0 ** -24, eps_needed   # the budget is below one fp32 rounding of F0: injection is a property of the problem


def test_kokoro_canonical_short_sentence_forced_durations(setup):
    """T = 80 tokens, durations forced to 264 frames -> 158 400 samples (6.6 s)."""
    S, eng, ref = setup
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, durs = eng.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                             overrides=_teacher(tr))
    torch.cuda.synchronize()
    assert torch.equal(durs[0].cpu(), fd)
    got = outs[0].cpu()
    assert got.numel() == 158400
    peak = float(audio_ref.abs().max())
    err = float((got - audio_ref[0]).abs().max())
    snr = snr_db(got, audio_ref[0])
    print(f"kokoro canonical (teacher-forced vocoder): peak={peak:.3f} max_abs_err={err:.3e} snr={snr:.1f} dB")
    assert err <= 2e-3 * peak and snr >= 50.0


def test_kokoro_batch_equals_single(setup):
    """Utterance batching is new (the reference is batch-1): every item of a ragged batch must reproduce
    its single-utterance result.  Not bitwise: conv_gemm picks its kernel by the number of 128 x 128 tiles of a launch (the batch fills the
    wave-specialised kernel from 128 tiles on, a lone short utterance runs the 4-wave one, launches of few tiles split their K loop over several
    workgroups), and the variants add in different orders; the ~1e-7 relative differences per layer reach 2e-5 of the waveform peak at the output
    (measured 1.7e-5).  The comparison is FREE-RUNNING (F0 is not injected), so it also depends on no harmonic phase feature sitting at its +-pi wrap
    for these inputs: a change of the split grouping moved F0 by 5e-6 relative and wrapped one (round 4, tools/diag_batch_single.py) -- the failure
    mode to look for first when this test breaks after a kernel-selection change.
    Mode 6 (the default since round 6): the batch's stage-1 launches reach 128 tiles and take the FP4 lo pass while a lone utterance's stay on the 4-wave
    kernels (fp16 hi + fp16 lo on the same image: MORE exact), so the two differ by the FP4 lo pass's own error -- measured 9.2e-5 of the peak; the bar
    is 4e-4 there (the mode's error against the oracle is 2.3e-4, bar 2e-3).  The exact modes keep 5e-5."""
    S, eng, _ = setup
    voice = S.make_voice_pack()
    idl = [S.make_phoneme_ids(n, seed=10 + n) for n in (12, 25, 7)]
    refs = torch.cat([voice[len(i) - 3] for i in idl], 0)
    fds = [S.forced_durations(len(i), 3 * len(i), seed=len(i)) for i in idl]
    Fm = max(int(f.sum()) for f in fds)
    rng = np.random.default_rng(5)
    ri = torch.from_numpy(rng.uniform(size=(3, 9)).astype(np.float32))
    nz = torch.from_numpy(rng.standard_normal((3, 2 * Fm * 300, 9)).astype(np.float32))
    outs, durs = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz)
    torch.cuda.synchronize()
    for b in range(3):
        Fb = int(fds[b].sum())
        o1, _ = eng.forward([idl[b]], refs[b:b + 1], forced_durations=[fds[b]], rand_ini=ri[b:b + 1],
                            noise=nz[b:b + 1, : 2 * Fb * 300].contiguous())
        torch.cuda.synchronize()
        assert outs[b].shape == o1[0].shape
        d = float((outs[b] - o1[0]).abs().max())
        assert d <= (4e-4 if eng.precision == 6 else 5e-5) * float(o1[0].abs().max() + 1), (b, d)


def test_kokoro_single_pass_bf16_precision_mode(setup):
    """precision=1 (one bf16 MFMA pass per product) is the fast mode; its error is reported, loosely bounded."""
    S, eng, ref = setup
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    eng1 = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=1)
    ids = S.make_phoneme_ids(10, seed=3)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(len(ids), 30)
    ri, nz = _noise(30, 9)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, _ = eng1.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                           overrides=_teacher(tr))
    torch.cuda.synchronize()
    snr = snr_db(outs[0].cpu(), audio_ref[0])
    print(f"kokoro precision=1 (single bf16 pass), teacher-forced vocoder: snr={snr:.1f} dB")
    assert snr >= 25.0


def test_kokoro_fp16_single_pass_precision_mode(setup):
    """precision=3 (opt-in fast mode): decoder / generator convs as ONE fp16 MFMA pass (activations rounded to fp16,
    bf16-valued weights held in fp16, fp32 accumulation); the front end stays on the hi+lo split, so the integer path and
    the F0 / N curves are bit-identical to the default engine.  On the canonical sentence it reaches SNR ~57 dB but a
    max-abs error of ~2.4e-3 * peak: it MISSES the default mode's 2e-3 * peak bar, which is why it
    is not the default and not the headline number.  Bounds asserted here: SNR >= 50 dB, max-abs <= 3e-3 * peak."""
    S, eng, ref = setup
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    eng3 = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=3)
    # free-running front end: identical integer path and F0 / N curves as the default engine
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    _, d3, t3 = eng3.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    _, d2, t2 = eng.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    torch.cuda.synchronize()
    assert torch.equal(d3[0], d2[0]) and torch.equal(t3["f0"], t2["f0"]) and torch.equal(t3["n"], t2["n"])
    # canonical sentence, vocoder teacher-forced on the oracle's features
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, _ = eng3.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                           overrides=_teacher(tr))
    torch.cuda.synchronize()
    got = outs[0].cpu()
    peak = float(audio_ref.abs().max())
    err = float((got - audio_ref[0]).abs().max())
    snr = snr_db(got, audio_ref[0])
    print(f"kokoro precision=3 (fp16 single pass, canonical sentence): peak={peak:.3f} max_abs_err={err:.3e} snr={snr:.1f} dB")
    assert err <= 3e-3 * peak and snr >= 50.0, (err, snr)


def test_kokoro_precision5_mx_lo_pass_mode(setup):
    """precision=5: decoder / generator convs as the fp16 hi pass + block-scaled e4m3 lo pass (MX images, v_mfma_scale_f32_32x32x64_f8f6f4) where the
    wave-specialised kernel takes the shape, fp16 hi + lo elsewhere; the front end stays on the bf16 hi+lo split (integer path and F0 / N curves
    bit-identical to the default engine).  Same bars as the default mode: 2e-3 * peak and 50 dB, teacher-forced on the oracle's features, on the
    canonical sentence -- as a batch of four so that BOTH generator stages fill the chip and run the MX kernel (one utterance leaves stage 0 on the
    4-wave kernels)."""
    S, eng, ref = setup
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    assert eng.precision == KokoroEngine.default_precision(torch.bfloat16) == 6, "the engine default for a bf16 checkpoint is the benchmarked mode (KokoroEngine.default_precision)"
    eng5 = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=5)
    eng2 = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=2)
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    _, d5, t5 = eng5.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    _, d2, t2 = eng2.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    torch.cuda.synchronize()
    assert torch.equal(d5[0], d2[0]) and torch.equal(t5["f0"], t2["f0"]) and torch.equal(t5["n"], t2["n"])
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    nb = 4
    tea = {k: torch.cat([torch.as_tensor(v)] * nb, dim=0) for k, v in _teacher(tr).items()}
    outs, _ = eng5.forward([ids] * nb, ref_s.repeat(nb, 1), forced_durations=[fd] * nb, rand_ini=torch.from_numpy(np.repeat(ri, nb, axis=0)),
                           noise=torch.from_numpy(np.repeat(nz, nb, axis=0)), overrides=tea)
    torch.cuda.synchronize()
    peak = float(audio_ref.abs().max())
    worst_err, worst_snr = 0.0, 1e9
    for b in range(nb):
        got = outs[b].cpu()
        worst_err = max(worst_err, float((got - audio_ref[0]).abs().max()))
        worst_snr = min(worst_snr, snr_db(got, audio_ref[0]))
    print(f"kokoro precision=5 (fp16 hi + MX e4m3 lo, canonical sentence x {nb}): peak={peak:.3f} max_abs_err={worst_err:.3e} ({worst_err / peak:.2e} of peak) snr={worst_snr:.1f} dB")
    assert worst_err <= 2e-3 * peak and worst_snr >= 50.0, (worst_err, worst_snr)
    # single utterance: stage 0 on the 4-wave kernels (precision-4 arithmetic on the same images), stage 1 on the MX kernel
    outs1, _ = eng5.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz), overrides=_teacher(tr))
    torch.cuda.synchronize()
    got = outs1[0].cpu()
    err1, snr1 = float((got - audio_ref[0]).abs().max()), snr_db(got, audio_ref[0])
    print(f"kokoro precision=5, one utterance: max_abs_err={err1:.3e} snr={snr1:.1f} dB")
    assert err1 <= 2e-3 * peak and snr1 >= 50.0, (err1, snr1)
    # the bf16 hi + lo mode (2) stays covered on the same sentence: same bars
    outs2, _ = eng2.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz), overrides=_teacher(tr))
    torch.cuda.synchronize()
    got = outs2[0].cpu()
    err2, snr2 = float((got - audio_ref[0]).abs().max()), snr_db(got, audio_ref[0])
    print(f"kokoro precision=2, one utterance: max_abs_err={err2:.3e} snr={snr2:.1f} dB")
    assert err2 <= 2e-3 * peak and snr2 >= 50.0, (err2, snr2)


def test_kokoro_precision6_fp4_lo_pass_mode(setup):
    """precision=6 (round 6): the >= 7-tap decoder / generator convs as the fp16 hi pass + block-scaled FP4 (e2m1) lo pass on MX4 images -- the lo pass
    at 4x the 16-bit rate of the matrix pipe.  The SAME bars as every mode (2e-3 * peak, 50 dB), teacher-forced on the oracle's features, canonical
    sentence x 4 (both generator stages on the wave-specialised kernel) and x 1; the front end (integer path, F0 / N curves) stays bit-identical to
    mode 2.  Expected from the CPU study of the scheme (profiles/r6_split_format_study_fp6.txt): ~4e-4 of the peak, ~70 dB."""
    S, eng, ref = setup
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    eng6 = eng if eng.precision == 6 else KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=6)
    eng2 = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=2)
    ids = S.make_phoneme_ids(18, seed=5)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    _, d6, t6 = eng6.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    _, d2, t2 = eng2.forward([ids], ref_s, speed=1.3, return_intermediates=True)
    torch.cuda.synchronize()
    assert torch.equal(d6[0], d2[0]) and torch.equal(t6["f0"], t2["f0"]) and torch.equal(t6["n"], t2["n"])
    del eng2
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    peak = float(audio_ref.abs().max())
    for nb in (4, 1):
        tea = {k: torch.cat([torch.as_tensor(v)] * nb, dim=0) for k, v in _teacher(tr).items()}
        outs, _ = eng6.forward([ids] * nb, ref_s.repeat(nb, 1), forced_durations=[fd] * nb, rand_ini=torch.from_numpy(np.repeat(ri, nb, axis=0)),
                               noise=torch.from_numpy(np.repeat(nz, nb, axis=0)), overrides=tea)
        torch.cuda.synchronize()
        worst_err = max(float((outs[b].cpu() - audio_ref[0]).abs().max()) for b in range(nb))
        worst_snr = min(snr_db(outs[b].cpu(), audio_ref[0]) for b in range(nb))
        print(f"kokoro precision=6 (fp16 hi + MX FP4 lo, canonical sentence x {nb}): peak={peak:.3f} max_abs_err={worst_err:.3e} ({worst_err / peak:.2e} of peak) snr={worst_snr:.1f} dB")
        assert worst_err <= 2e-3 * peak and worst_snr >= 50.0, (nb, worst_err, worst_snr)


def test_kokoro_default_mode_batch64_canonical(setup):
    """The BENCHMARKED configuration is the parity-tested one: 64 canonical utterances (T = 80, F = 264) through bench.py's exact call path --
    ``shard.kokoro_step`` on a ``ShardChannel`` (world 1), the engine in its default mode (6 for a bf16 checkpoint since round 6), ``back_kwargs`` carrying the
    SineGen inputs -- teacher-forced on the oracle's F0 / N / harmonic features, EVERY one of the 64 waveforms against the fp32 oracle at the
    2e-3 * peak / 50 dB bars.  At this launch size the generator runs conv_ws4_kernel<5, 2, ...> x 36, <2, 2> x 12, <2, 1> x 21 per pass
    (profiles/r4_kernel_stats_b64_call21.txt): a different instantiation mix from the 4-utterance test above."""
    S, eng, ref = setup
    from mlx_audio_amd import shard

    assert eng.precision == 6   # KokoroEngine.default_precision for a bf16 checkpoint (round 6: the FP4 lo pass)
    nb = 64
    ids = S.make_phoneme_ids(78)
    voice = S.make_voice_pack()
    ref_s = voice[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    dev = eng.dev
    tea = {k: torch.as_tensor(v).to(dev).expand(nb, *torch.as_tensor(v).shape[1:]).contiguous() for k, v in _teacher(tr).items()}
    ri_d = torch.from_numpy(ri).to(dev).expand(nb, -1).contiguous()
    nz_d = torch.from_numpy(nz).to(dev).expand(nb, -1, -1).contiguous()
    fd_d = fd.to(dev)
    ch = shard.ShardChannel(dev, None, max_items=nb, max_tokens=512)
    outs = shard.kokoro_step(ch, eng, [ids] * nb, lambda i, t: voice[t - 3].to(dev), 600, forced_durations_of=lambda i: fd_d,
                             back_kwargs=lambda items: dict(rand_ini=ri_d[: len(items)], noise=nz_d[: len(items)],
                                                            overrides={k: v[: len(items)] for k, v in tea.items()}))
    torch.cuda.synchronize()
    assert len(outs) == nb
    peak = float(audio_ref.abs().max())
    got = torch.stack([o.cpu() for o in outs])
    err = (got - audio_ref[0][None]).abs().amax(dim=1)
    snrs = [snr_db(got[b], audio_ref[0]) for b in range(nb)]
    print(f"kokoro default mode ({eng.precision}), canonical sentence x {nb} through shard.kokoro_step: peak={peak:.3f} worst max_abs_err={float(err.max()):.3e} "
          f"({float(err.max()) / peak:.2e} of peak) worst snr={min(snrs):.1f} dB; spread over the batch {float((got - got[0:1]).abs().max()):.2e}")
    assert float(err.max()) <= 2e-3 * peak and min(snrs) >= 50.0, (float(err.max()), min(snrs))


def test_kokoro_identical_utterances_identical_waveforms(setup):
    """Regression (round 5): 16 IDENTICAL canonical utterances in one free-running call (same ids, style, durations, SineGen inputs; 192 tiles per
    predictor conv: the wave-specialised kernels) -- every row walks the same tiles of the same kernels, so the 16 F0 / N curves and the 16
    waveforms must be bit-identical, and a second call must reproduce them.  (The fused instance-norm statistics used to come back with random
    errors at >= 128 tiles: F0 differed by 1e-2 between rows and runs, tools/diag_batch_rows.py.)"""
    S, eng, _ = setup
    B = 16
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    ri, nz = _noise(264, 1234)
    ri_d = torch.from_numpy(ri).cuda().expand(B, -1).contiguous()
    nz_d = torch.from_numpy(nz).cuda().expand(B, -1, -1).contiguous()
    prev = None
    for rep in range(2):
        outs, _, tr = eng.forward([ids] * B, ref_s.repeat(B, 1), forced_durations=[fd] * B, rand_ini=ri_d, noise=nz_d, return_intermediates=True)
        torch.cuda.synchronize()
        wav = torch.stack(outs)
        for k in ("f0", "n", "xg", "stage0", "stage1"):
            assert torch.equal(tr[k], tr[k][0:1].expand_as(tr[k])), (rep, k, float((tr[k] - tr[k][0:1]).abs().max()))
        assert torch.equal(wav, wav[0:1].expand_as(wav)), (rep, float((wav - wav[0:1]).abs().max()))
        if prev is not None:
            assert torch.equal(wav, prev), float((wav - prev).abs().max())
        prev = wav
