"""Edge cases of the hot path on the GPU: the smallest and the largest inputs the reference accepts, ragged batches with extreme length ratios, empty audio.

The oracle comparison is used where the oracle finishes in seconds (tiny inputs); at the maximum sizes the checks are size-independent properties
(a ragged batch reproduces its single-utterance results, sample counts = 600 x frames, outputs finite), as the task statement prescribes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_kokoro_gpu import _noise, _teacher, setup, snr_db  # noqa: E402,F401  (module-scoped fixture + helpers)


def test_kokoro_smallest_utterance_against_oracle(setup):
    """One phoneme between BOS / EOS (T = 3), one frame per token (F = 3 -> 1 800 samples): every kernel at its minimum shape (single-row tiles,
    reflect padding longer than the signal's interior, 1-step LSTMs)."""
    S, eng, ref = setup
    ids = torch.tensor([0, 57, 0])
    ref_s = S.make_voice_pack()[0]
    fd = torch.tensor([1, 1, 1], dtype=torch.int32)
    ri, nz = _noise(3, 9)
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    outs, durs, tg = eng.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz),
                                 overrides=_teacher(tr), return_intermediates=True)
    torch.cuda.synchronize()
    assert torch.equal(durs[0].cpu(), fd) and outs[0].numel() == 1800
    got = outs[0].cpu()
    peak = float(audio_ref.abs().max())
    assert float((got - audio_ref[0]).abs().max()) <= 2e-3 * peak
    assert snr_db(got, audio_ref[0]) >= 50.0
    assert float((tg["f0"][0].cpu() - tr["f0"][0]).abs().max() / tr["f0"][0].abs().max()) < 5e-4


def test_kokoro_maximum_tokens_in_a_ragged_batch(setup):
    """510 phonemes (T = 512 = max_position_embeddings, the reference's chunk limit, pipeline.py:266-293) next to a 2-phoneme utterance: the long item
    must not disturb the short one and vice versa, sample counts are 600 x frames.  A 4-frame utterance normalises its F0 / N predictor features over
    4-8 samples (AdaIN statistics), which amplifies the ~1e-6 kernel-choice differences between a batch and a single run to ~1e-4 of F0 -- and
    SineGen turns any F0 difference into a different phase (tests/test_kokoro_gpu.py) -- so the waveform comparison injects each single run's
    F0 / N into the batch, and the free-running curves are held to the front end's own bar (5e-4)."""
    S, eng, _ = setup
    voice = S.make_voice_pack()
    idl = [S.make_phoneme_ids(510, seed=3), S.make_phoneme_ids(2, seed=4)]
    assert idl[0].numel() == 512
    refs = torch.cat([voice[len(i) - 3] for i in idl], 0)
    fds = [torch.ones(len(i), dtype=torch.int32) for i in idl]      # one frame per token keeps the run short: F = 512 and 4
    fds[0][100:110] = 3                                              # ... with a few longer ones: F = 532
    Fs = [int(f.sum()) for f in fds]
    Fm = max(Fs)
    rng = np.random.default_rng(11)
    ri = torch.from_numpy(rng.uniform(size=(2, 9)).astype(np.float32))
    nz = torch.from_numpy(rng.standard_normal((2, 2 * Fm * 300, 9)).astype(np.float32))
    singles, f0s, ns = [], [], []
    for b in range(2):
        o1, _, t1 = eng.forward([idl[b]], refs[b:b + 1], forced_durations=[fds[b]], rand_ini=ri[b:b + 1], noise=nz[b:b + 1, : 2 * Fs[b] * 300].contiguous(),
                                return_intermediates=True)
        torch.cuda.synchronize()
        singles.append(o1[0])
        f0s.append(t1["f0"][0])
        ns.append(t1["n"][0])
    _, _, tb = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz, return_intermediates=True)
    for b in range(2):   # free-running pitch / energy curves of the batch against the single runs
        for got, want in ((tb["f0"][b, : 2 * Fs[b]], f0s[b]), (tb["n"][b, : 2 * Fs[b]], ns[b])):
            assert float((got - want).abs().max() / want.abs().max()) < 5e-4
    f0 = torch.zeros(2, 2 * Fm, device=f0s[0].device)
    n = torch.zeros(2, 2 * Fm, device=f0s[0].device)
    for b in range(2):
        f0[b, : 2 * Fs[b]], n[b, : 2 * Fs[b]] = f0s[b], ns[b]
    outs, durs = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz, overrides=dict(f0=f0, n=n))
    torch.cuda.synchronize()
    for b in range(2):
        assert outs[b].numel() == 600 * Fs[b] and bool(torch.isfinite(outs[b]).all()) and torch.equal(durs[b].cpu(), fds[b])
        d = float((outs[b] - singles[b]).abs().max())
        # not bitwise: the single runs take conv_gemm's split-K path (few output tiles), the batch partly does not, so the fp32 accumulation order
        # differs conv by conv; measured 3.3e-4 on a peak of 2.06 (1.6e-4 of peak; the oracle bar of the same waveform is 2e-3 of peak)
        assert d <= 2e-4 * float(singles[b].abs().max() + 1), (b, d)
    with pytest.raises(AssertionError):   # one token more than the position table holds
        eng.forward([torch.cat([idl[0], torch.zeros(1, dtype=torch.long)])], refs[:1])


def test_kokoro_front_back_split_equals_forward(setup):
    """forward == back(front(...)), also after the state went through the wire format of a re-balance (pack -> unpack on a 'different GPU')."""
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroFront

    S, eng, _ = setup
    voice = S.make_voice_pack()
    idl = [S.make_phoneme_ids(n, seed=20 + n) for n in (9, 17)]
    refs = torch.cat([voice[len(i) - 3] for i in idl], 0)
    fds = [S.forced_durations(len(i), 3 * len(i), seed=len(i)) for i in idl]
    Fm = max(int(f.sum()) for f in fds)
    rng = np.random.default_rng(6)
    ri = torch.from_numpy(rng.uniform(size=(2, 9)).astype(np.float32))
    nz = torch.from_numpy(rng.standard_normal((2, 2 * Fm * 300, 9)).astype(np.float32))
    want, _ = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz)
    st = eng.front(idl, refs, forced_durations=fds)
    width, style = eng.hid + eng.sty, 2 * eng.sty
    blobs = [st.pack(i) for i in range(2)]
    assert [int(b.numel()) for b in blobs] == [KokoroFront.packed_size(len(i), style, width) for i in idl]
    moved = KokoroFront.unpack(blobs, st.frames, style, width, st.speed)
    got, _ = eng.back(moved, rand_ini=ri, noise=nz)
    torch.cuda.synchronize()
    for b in range(2):
        assert torch.equal(got[b], want[b])    # same kernels on the same values: bit-identical


def test_whisper_short_and_empty_audio():
    """generate() on half a second of audio (one partly filled window) and on NO audio (whisper.py:1026: the window loop does not run)."""
    from mlx_audio_amd.stt.models.whisper import Model
    from mlx_audio_amd.stt.models.whisper import synthetic as WS

    from mlx_audio_amd.stt.models.whisper import ModelDimensions

    # generate() works on 30 s windows of 3000 mel frames, so the encoder context must be the real 1500; everything else is small
    dims = ModelDimensions(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_vocab=51865, n_text_ctx=64,
                           n_text_state=128, n_text_head=2, n_text_layer=1)
    m = Model(dims, device="cuda")
    m.load_weights(m.sanitize(WS.make_whisper_weights(dims, seed=1)))
    rng = np.random.default_rng(0)
    out = m.generate(rng.standard_normal(8000).astype(np.float32) * 0.1, language="en", temperature=0.0, sample_len=6)
    assert isinstance(out.text, str) and all(s["seek"] == 0 for s in out.segments)
    empty = m.generate(np.zeros(0, np.float32), language="en", temperature=0.0, sample_len=6)
    assert empty.text == "" and empty.segments == []


def test_conv_gemm_batch_with_an_empty_item():
    """A ragged batch in which one item has NO valid rows (lens = 0): its output rows stay untouched, the other items are exact."""
    from mlx_audio_amd import ops

    g = torch.Generator().manual_seed(2)
    B, L, Cin, Cout, K = 3, 300, 64, 128, 3
    x = torch.randn(B, L, Cin, generator=g)
    w = (torch.randn(Cout, K, Cin, generator=g) / (K * Cin) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(Cout, generator=g) * 0.1
    lens = torch.tensor([300, 0, 77], dtype=torch.int32)
    pc = ops.pack_conv(w, b, "cuda")
    y = torch.full((B, L, Cout), 7.0, device="cuda")
    ops.conv_gemm(x.cuda(), pc, y, pad=1, lens_in=lens.cuda(), lens_out=lens.cuda())
    torch.cuda.synchronize()
    assert float((y[1] - 7.0).abs().max()) == 0.0
    for i in (0, 2):
        n = int(lens[i])
        xi = torch.nn.functional.pad(x[i, :n].double().t()[None], (1, 1))
        want = torch.nn.functional.conv1d(xi, w.double().permute(0, 2, 1), b.double())[0].t()
        got = y[i, :n].cpu().double()
        assert float((got - want).abs().max() / want.abs().max()) < 3e-5
        assert float((y[i, n:] - 7.0).abs().max()) == 0.0 if n < L else True
