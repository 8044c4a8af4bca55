"""Whisper (BASELINE config[2]) on the HIP path vs the CPU oracle.

Tolerances (the reference itself computes Whisper in fp16, i.e. with 2^-11 relative rounding per op; its own numeric tests
use rtol = atol = 2e-3):
  * encoder / decoder activations, precision 4 (fp16 hi+lo activations, f32 attention): <= 3e-4 relative to the tensor peak;
    precision 3 (single fp16 pass = the reference's own activation rounding): <= 1e-2
  * logits, teacher-forced on the oracle's tokens: <= 2e-3 * peak
  * integer path (tokens): bit-exact wherever the oracle's top-2 margin of the filtered logits exceeds 10x the measured logit
    error (margin asserted per step); free-running tokens bit-exact under the same condition
  * sum_logprobs: 1e-3 absolute per step
Needs a real MI355X: ``pytest -m gpu``.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _margin as margin_rule  # noqa: E402  (tests/_margin.py)
DEV = "cuda"


def rel_peak(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    from mlx_audio_amd import ops
    from mlx_audio_amd.stt.models.whisper import synthetic as WS
    from mlx_audio_amd.stt.models.whisper.engine import WhisperEngine
    from oracle.whisper_ref import TokenizerSpec, WhisperRef

    ops.require_gpu()
    dims = WS.tiny_dims()
    w = WS.make_whisper_weights(dims, seed=1)
    return dict(dims=dims, eng=WhisperEngine(w, dims, device=DEV), eng3=WhisperEngine(w, dims, device=DEV, precision=3),
                ref=WhisperRef(w, dims), tok=TokenizerSpec(), WS=WS)


def test_tiny_encoder_layers(tiny):
    mel = tiny["WS"].make_mel(2, seed=4, n_frames=2 * tiny["dims"].n_audio_ctx)
    exp, exp_layers = tiny["ref"].encoder(mel, return_layers=True)
    got, layers = tiny["eng"].encode(mel, return_layers=True)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(layers, exp_layers)):
        assert rel_peak(a, b) < 3e-4, (i, rel_peak(a, b))
    assert rel_peak(got, exp) < 3e-4
    got3 = tiny["eng3"].encode(mel)
    assert rel_peak(got3, exp) < 1e-2


def test_encoder_on_pre_split_activations_equals_the_float_path(tiny):
    """engine.split_acts (the default at precision 4): LayerNorm / GELU epilogue / one pass behind the attention leave fp16 hi | lo words and the linears
    read those -- the same two numbers per value the float path's prologue makes, so only the kernel choice (and with it the accumulation order) may differ."""
    eng = tiny["eng"]
    assert eng.split_acts
    mel = tiny["WS"].make_mel(3, seed=5, n_frames=2 * tiny["dims"].n_audio_ctx)
    a = eng.encode(mel).clone()
    eng.split_acts = False
    try:
        b = eng.encode(mel).clone()
    finally:
        eng.split_acts = True
    torch.cuda.synchronize()
    assert rel_peak(a, b) < 1e-5, rel_peak(a, b)   # (measured 2.4e-6: fp32 accumulation order over four layers)


def _margin(filtered):
    top2 = torch.topk(filtered, 2, dim=-1).values
    return (top2[:, 0] - top2[:, 1])


@pytest.mark.parametrize("without_timestamps", [False, True])
def test_tiny_decode_teacher_forced(tiny, without_timestamps):
    dims, tok = tiny["dims"], tiny["tok"]
    mel = tiny["WS"].make_mel(2, seed=5, n_frames=2 * dims.n_audio_ctx)
    suppress = [1, 2, 3, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.transcribe, tok.translate]
    kw = dict(sample_len=12, without_timestamps=without_timestamps, suppress_tokens=suppress)
    free = tiny["ref"].decode(mel, tok, **kw, record=True)
    sb = free["sample_begin"]
    steps = free["tokens"].shape[1] - sb
    forced = free["tokens"][:, sb:]
    exp = tiny["ref"].decode(mel, tok, **kw, forced_tokens=forced, record=True)
    got = tiny["eng"].decode(mel, tok, **kw, forced_tokens=forced[:, :steps], record=True)
    torch.cuda.synchronize()
    assert rel_peak(got["audio_features"], exp["audio_features"]) < 3e-4
    np.testing.assert_allclose(got["no_speech_probs"].cpu().numpy(), exp["no_speech_probs"].numpy(), rtol=2e-3, atol=1e-7)
    assert torch.equal(got["tokens"].cpu(), exp["tokens"][:, :got["tokens"].shape[1]])
    for i in range(steps):
        e, g = exp["trace"][i], got["trace"][i]
        err = float((g["raw"].cpu() - e["raw"]).abs().max())
        peak = float(e["raw"].abs().max())
        assert err <= 2e-3 * peak, (i, err, peak)
        assert torch.equal(torch.isinf(g["filtered"].cpu()), torch.isinf(e["filtered"])), f"step {i}: mask pattern differs"
        # integer path: wherever the oracle's decision margin is clear of the logit error the arg-max must agree
        m = _margin(e["filtered"])
        clear = m > 10 * err
        if bool(clear.any()):
            assert torch.equal(g["filtered"].cpu().argmax(-1)[clear], e["filtered"].argmax(-1)[clear]), f"step {i}"
    np.testing.assert_allclose(got["sum_logprobs"].cpu().numpy(), exp["sum_logprobs"].numpy(), atol=1e-3 * steps)


def test_tiny_decode_free_running(tiny):
    dims, tok = tiny["dims"], tiny["tok"]
    mel = tiny["WS"].make_mel(3, seed=6, n_frames=2 * dims.n_audio_ctx)
    kw = dict(sample_len=10, suppress_tokens=[tok.sot, tok.no_speech])
    exp = tiny["ref"].decode(mel, tok, **kw, record=True)
    got = tiny["eng"].decode(mel, tok, **kw, poll=4)
    torch.cuda.synchronize()
    margins = torch.stack([_margin(t["filtered"]) for t in exp["trace"]], dim=1)  # [B, steps]
    gt, et = got["tokens"].cpu(), exp["tokens"]
    sb = exp["sample_begin"]
    for b in range(et.shape[0]):
        n = et.shape[1]
        # beyond a knife-edge decision two fp32 builds may diverge (tests/_margin.py keeps count of what that hides)
        done = margin_rule.walk("whisper", gt[b, sb:n].tolist(), et[b, sb:n].tolist(), margins[b].tolist(), where=("free", b))
        if done == n - sb:
            assert abs(float(got["sum_logprobs"][b]) - float(exp["sum_logprobs"][b])) < 1e-2


def test_tiny_decode_prompt_conditioned(tiny):
    """decoding.py:525-551: ``[sot_prev] + prompt + sot_sequence`` as the initial sequence; sample_begin / sot_index move with it.  Teacher-forced
    so that the per-step filtered logits are compared on identical contexts (fp32 bar as in the unconditioned test)."""
    dims, tok = tiny["dims"], tiny["tok"]
    mel = tiny["WS"].make_mel(2, seed=15, n_frames=2 * dims.n_audio_ctx)
    initial = [tok.sot_prev] + [11, 12, 13, 14, 15, 16, 17] + list(tok.sot_sequence)
    steps = 6
    g = torch.Generator().manual_seed(3)
    forced = torch.randint(0, tok.timestamp_begin - 200, (2, steps), generator=g)
    kw = dict(sample_len=steps, suppress_tokens=[tok.sot, tok.no_speech], initial_tokens=initial, forced_tokens=forced, record=True)
    exp = tiny["ref"].decode(mel, tok, **kw)
    got = tiny["eng"].decode(mel, tok, **kw)
    assert got["sample_begin"] == exp["sample_begin"] == len(initial)
    assert torch.equal(got["tokens"].cpu(), exp["tokens"])
    np.testing.assert_allclose(got["no_speech_probs"].cpu().numpy(), exp["no_speech_probs"].numpy(), rtol=2e-3, atol=1e-6)
    for gt, et in zip(got["trace"], exp["trace"]):
        assert rel_peak(gt["raw"], et["raw"]) < 2e-4
        fin = torch.isfinite(et["filtered"])
        assert torch.equal(torch.isfinite(gt["filtered"]).cpu(), fin)
    np.testing.assert_allclose(got["sum_logprobs"].cpu().numpy(), exp["sum_logprobs"].numpy(), atol=1e-3 * steps)


def test_tiny_decode_sampling_same_noise(tiny):
    """temperature > 0 (decoding.py:266-269) = arg-max of logits / T + Gumbel noise.  The engine draws the noise from a device generator; the
    test re-draws the same stream and hands it to the oracle, so the sampled tokens must agree wherever the noisy decision is not a knife edge."""
    dims, tok, eng = tiny["dims"], tiny["tok"], tiny["eng"]
    mel = tiny["WS"].make_mel(2, seed=16, n_frames=2 * dims.n_audio_ctx)
    T = 0.8
    vp = (dims.n_vocab + 3) // 4 * 4
    kw = dict(sample_len=8, suppress_tokens=[tok.sot, tok.no_speech], temperature=T)
    got = eng.decode(mel, tok, **kw, generator=torch.Generator(device=DEV).manual_seed(77), fixed_steps=True)
    g2 = torch.Generator(device=DEV).manual_seed(77)
    noise = []
    for _ in range(8):
        u = torch.rand((2, vp), dtype=torch.float32, device=DEV, generator=g2).clamp_(1e-20, 1.0 - 1e-7)
        noise.append((-(-u.log()).log()).cpu()[:, :dims.n_vocab])
    exp = tiny["ref"].decode(mel, tok, **kw, gumbel=lambda i: noise[i], record=True)
    gt, et = got["tokens"].cpu(), exp["tokens"]
    n = min(gt.shape[1], et.shape[1])
    sb = exp["sample_begin"]
    for b in range(2):
        m = [float(_margin(tr["filtered"][b:b + 1] / T + noise[i][b:b + 1])) for i, tr in enumerate(exp["trace"])][: n - sb]
        margin_rule.walk("whisper", gt[b, sb:n].tolist(), et[b, sb:n].tolist(), m, where=("sampled", b))
    assert not torch.equal(gt[:, sb:n], eng.decode(mel, tok, sample_len=8, suppress_tokens=[tok.sot, tok.no_speech],
                                                   fixed_steps=True)["tokens"].cpu()[:, sb:n])  # the noise changed something
    again = eng.decode(mel, tok, **kw, generator=torch.Generator(device=DEV).manual_seed(77), fixed_steps=True)
    assert torch.equal(again["tokens"], got["tokens"])  # seeded => reproducible


def test_batch_equals_single(tiny):
    """Size-independent property: a window decoded in a batch equals the same window decoded alone (bit-level on tokens)."""
    dims, tok = tiny["dims"], tiny["tok"]
    mel = tiny["WS"].make_mel(3, seed=7, n_frames=2 * dims.n_audio_ctx)
    kw = dict(sample_len=8, suppress_tokens=[tok.sot], fixed_steps=True)
    full = tiny["eng"].decode(mel, tok, **kw)
    for b in range(3):
        one = tiny["eng"].decode(mel[b:b + 1], tok, **kw)
        assert rel_peak(one["audio_features"][0], full["audio_features"][b]) < 1e-5
        assert torch.equal(one["tokens"][0].cpu(), full["tokens"][b].cpu())


def test_kv_cache_equals_full_context(tiny):
    """Incremental decoding through the KV cache == one prefill over the whole context (whisper.py:476-498)."""
    eng, dims = tiny["eng"], tiny["dims"]
    mel = tiny["WS"].make_mel(1, seed=8, n_frames=2 * dims.n_audio_ctx)
    xa = eng.encode(mel)
    toks = torch.tensor([[50258, 50259, 50359, 50364, 11, 22, 33, 44, 55, 66]], dtype=torch.int32, device=DEV)
    st = eng.new_state(xa)
    full = eng.logits(eng.decoder_step(toks, st))[:, :, :dims.n_vocab]       # flash (prefill) path, 10 queries
    st = eng.new_state(xa)
    parts = [eng.logits(eng.decoder_step(toks[:, :3], st))[:, :, :dims.n_vocab]]
    for i in range(3, toks.shape[1]):
        parts.append(eng.logits(eng.decoder_step(toks[:, i:i + 1], st))[:, :, :dims.n_vocab])  # gemv + decode-attention path
    torch.cuda.synchronize()
    inc = torch.cat(parts, dim=1)
    assert rel_peak(inc, full) < 2e-5
    # the native step runner (default) against the per-op Python schedule of the same kernels
    eng.native_decode = False
    try:
        st = eng.new_state(xa)
        py = [eng.logits(eng.decoder_step(toks[:, :3], st))[:, :, :dims.n_vocab]]
        for i in range(3, toks.shape[1]):
            py.append(eng.logits(eng.decoder_step(toks[:, i:i + 1], st))[:, :, :dims.n_vocab])
        torch.cuda.synchronize()
    finally:
        eng.native_decode = True
    assert rel_peak(torch.cat(py, dim=1), inc) < 1e-6


@pytest.mark.parametrize("B,rows_min", [(12, 9), (6, 5)])
def test_tiny_decode_tall_batch_on_the_rows_pipeline(tiny, B, rows_min):
    """Steps of 9..64 windows (and, with ``rows_min`` lowered, of 5..8) run the decoder on the rows pipeline -- tile images x input planes, the
    cross-attention block included (stack_step.cpp tall_step) -- teacher-forced against the oracle like the <= 8-window test, and against the same
    windows decoded in groups of <= 4 through the one-row kernels."""
    from mlx_audio_amd.stt.models.whisper.engine import WhisperEngine

    dims, tok = tiny["dims"], tiny["tok"]
    eng = tiny["eng"]
    if rows_min != eng.rows_min:
        eng = WhisperEngine(tiny["WS"].make_whisper_weights(dims, seed=1), dims, device=DEV)
        eng.rows_min = rows_min
    mel = tiny["WS"].make_mel(B, seed=15, n_frames=2 * dims.n_audio_ctx)
    suppress = [1, 2, 3, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.transcribe, tok.translate]
    kw = dict(sample_len=8, suppress_tokens=suppress)
    free = tiny["ref"].decode(mel, tok, **kw, record=True)
    sb = free["sample_begin"]
    forced = free["tokens"][:, sb:]
    steps = forced.shape[1]
    exp = tiny["ref"].decode(mel, tok, **kw, forced_tokens=forced, record=True)
    got = eng.decode(mel, tok, **kw, forced_tokens=forced[:, :steps], record=True)
    torch.cuda.synchronize()
    assert eng._rows_ws is not None                     # the tall path ran (it allocates the rows workspace)
    assert torch.equal(got["tokens"].cpu(), exp["tokens"][:, :got["tokens"].shape[1]])
    for i in range(steps):
        e, g = exp["trace"][i], got["trace"][i]
        err = float((g["raw"].cpu() - e["raw"]).abs().max())
        peak = float(e["raw"].abs().max())
        assert err <= 2e-3 * peak, (i, err, peak)
        m = _margin(e["filtered"])
        clear = m > 10 * err
        if bool(clear.any()):
            assert torch.equal(g["filtered"].cpu().argmax(-1)[clear], e["filtered"].argmax(-1)[clear]), f"step {i}"
    np.testing.assert_allclose(got["sum_logprobs"].cpu().numpy(), exp["sum_logprobs"].numpy(), atol=1e-3 * steps)
    # the same windows four at a time on the short path: logits of the last step agree to fp32 rounding of a different summation order
    for b0 in range(0, B, 4):
        part = tiny["eng"].decode(mel[b0:b0 + 4], tok, **kw, forced_tokens=forced[b0:b0 + 4, :steps], record=True)
        torch.cuda.synchronize()
        assert rel_peak(part["trace"][steps - 1]["raw"], got["trace"][steps - 1]["raw"][b0:b0 + 4]) < 2e-5


def test_whisper_small_tall_batch_full_size():
    """Whisper-small widths, 16 windows per step: three teacher-forced decode steps on the rows pipeline against the oracle."""
    from mlx_audio_amd.stt.models.whisper import synthetic as WS
    from mlx_audio_amd.stt.models.whisper.engine import WhisperEngine
    from oracle.whisper_ref import TokenizerSpec, WhisperRef

    dims = WS.WHISPER_SMALL
    w = WS.make_whisper_weights(dims, seed=0)
    eng = WhisperEngine(w, dims, device=DEV)
    ref = WhisperRef(w, dims)
    tok = TokenizerSpec()
    mel1 = WS.make_mel(2, seed=9)
    mel = mel1.repeat(8, 1, 1)                          # 16 windows: two distinct ones (the oracle encodes two)
    kw = dict(sample_len=3, suppress_tokens=[tok.sot, tok.no_speech])
    free = ref.decode(mel1, tok, **kw, record=True)
    forced = free["tokens"][:, free["sample_begin"]:]
    steps = forced.shape[1]
    got = eng.decode(mel, tok, **kw, forced_tokens=forced.repeat(8, 1), record=True)
    torch.cuda.synchronize()
    assert eng._rows_ws is not None
    for i in range(steps):
        e, g = free["trace"][i], got["trace"][i]
        for r in range(16):
            err = float((g["raw"][r].cpu() - e["raw"][r % 2]).abs().max())
            assert err <= 2e-3 * float(e["raw"].abs().max()), (i, r, err)


def test_whisper_small_full_size():
    """BASELINE config[2] at full size (whisper-small dims, one 30 s window): encoder output and 6 teacher-forced decode
    steps against the oracle."""
    from mlx_audio_amd.stt.models.whisper import synthetic as WS
    from mlx_audio_amd.stt.models.whisper.engine import WhisperEngine
    from oracle.whisper_ref import TokenizerSpec, WhisperRef

    dims = WS.WHISPER_SMALL
    w = WS.make_whisper_weights(dims, seed=0)
    eng = WhisperEngine(w, dims, device=DEV)
    ref = WhisperRef(w, dims)
    tok = TokenizerSpec()
    mel = WS.make_mel(1, seed=9)
    kw = dict(sample_len=6, suppress_tokens=[tok.sot, tok.no_speech])
    free = ref.decode(mel, tok, **kw, record=True)
    forced = free["tokens"][:, free["sample_begin"]:]
    steps = forced.shape[1]
    got = eng.decode(mel, tok, **kw, forced_tokens=forced, record=True)
    torch.cuda.synchronize()
    assert rel_peak(got["audio_features"], free["audio_features"]) < 5e-4
    for i in range(steps):
        e, g = free["trace"][i], got["trace"][i]
        err = float((g["raw"].cpu() - e["raw"]).abs().max())
        assert err <= 2e-3 * float(e["raw"].abs().max()), (i, err)
        m = _margin(e["filtered"])
        clear = m > 10 * err
        if bool(clear.any()):
            assert torch.equal(g["filtered"].cpu().argmax(-1)[clear], e["filtered"].argmax(-1)[clear])


def test_log_mel_on_speech_fixture(tiny):
    """The product front end (file -> ``load_audio`` -> fused STFT/power/mel/log10 kernel, audio.py:41-82) on real speech: the committed
    16 kHz clip of a wav the reference ships (tests/golden/make_wav_fixture.py), with generate()'s 30 s of padding, against the all-float64
    evaluation of the same chain.  Bar 1e-4 on a scale of [-0.78, 1.22] (the float32 restatement itself is within 8e-6)."""
    import os

    from mlx_audio_amd.stt.models.whisper import Model
    from mlx_audio_amd.stt.models.whisper.audio import N_FRAMES, N_SAMPLES
    from mlx_audio_amd.stt.utils import load_audio
    from oracle import dsp_ref

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "genesis_1_1_af_heart_16k.wav")
    model = Model(tiny["dims"], device=DEV)
    mel, content_frames = model._prepare_audio(path)
    torch.cuda.synchronize()
    a = load_audio(path)
    assert a.dtype == np.float32 and a.shape == (105600,) and float(np.abs(a).max()) <= 1.0
    ref = dsp_ref.whisper_log_mel_f64(a, padding=N_SAMPLES)
    assert tuple(mel.shape) == ref.shape == (3660, 80) and content_frames == 3660 - N_FRAMES == 660
    err = np.abs(mel.cpu().numpy().astype(np.float64) - ref)
    print(f"log-mel on speech fixture: max |err| vs float64 = {err.max():.2e}, mean = {err.mean():.2e}")
    assert err.max() < 1e-4


def test_model_surface(tiny):
    """The reference-shaped API: Model(dims) / load_weights / embed_audio / logits / decode / generate -> STTOutput."""
    from mlx_audio_amd.stt.models.whisper import Model
    from mlx_audio_amd.stt.models.whisper.decoding import DecodingOptions

    dims, WS = tiny["dims"], tiny["WS"]
    model = Model(dims, device=DEV)
    w = WS.make_whisper_weights(dims, seed=1)
    model.load_weights(model.sanitize(w))
    mel = WS.make_mel(1, seed=4, n_frames=2 * dims.n_audio_ctx)[0]
    feats = model.embed_audio(mel)
    assert tuple(feats.shape) == (dims.n_audio_ctx, dims.n_audio_state)
    lg = model.logits(torch.tensor([[50258, 50259, 50359]]), feats[None])
    assert tuple(lg.shape) == (1, 3, dims.n_vocab)
    res = model.decode(mel, DecodingOptions(language="en", sample_len=5, suppress_tokens=(50258,)))
    assert isinstance(res.tokens, list) and len(res.tokens) <= 5 and np.isfinite(res.avg_logprob) and 0.0 <= res.no_speech_prob <= 1.0
    with pytest.raises(NotImplementedError):
        model.decode(mel, DecodingOptions(beam_size=2))
    # best-of-n sampling groups (decoding.py:655-690) and prompt / prefix conditioning go through the same device loop
    gen = torch.Generator(device=DEV).manual_seed(5)
    res = model.decode(mel, DecodingOptions(language="en", sample_len=5, temperature=0.7, best_of=3, prompt=[7, 8, 9], prefix=[21, 22]),
                       generator=gen)
    assert len(res.tokens) <= 5 and res.temperature == 0.7 and np.isfinite(res.avg_logprob)  # the prefix sits before sample_begin
    with pytest.raises(ValueError):
        model.decode(mel, DecodingOptions(beam_size=2, best_of=2))


def test_detect_language_contract(tiny):
    """decoding.py:20-77: (language tokens, {code: probability}) from one decoder pass on <|startoftranscript|> with everything but the language tokens
    masked: the probabilities are the soft-max over the language columns of the oracle's logits, the token is their arg-max; single clip -> scalars."""
    from mlx_audio_amd.stt.models.whisper import Model

    dims, WS = tiny["dims"], tiny["WS"]
    model = Model(dims, device=DEV)
    model.load_weights(model.sanitize(WS.make_whisper_weights(dims, seed=1)))
    mel = WS.make_mel(2, seed=9, n_frames=2 * dims.n_audio_ctx)
    tokens, probs = model.detect_language(mel)
    tok = model.get_tokenizer()
    ids = list(tok.all_language_tokens)
    feats = tiny["ref"].encoder(mel)
    ref_logits = tiny["ref"].decoder(torch.full((2, 1), tok.sot, dtype=torch.long), feats)[0][:, 0, ids].double()
    want = torch.softmax(ref_logits, dim=-1)
    assert tokens.shape == (2,) and len(probs) == 2 and list(probs[0]) == list(tok.all_language_codes)
    for b in range(2):
        got = torch.tensor([probs[b][c] for c in tok.all_language_codes], dtype=torch.float64)
        assert abs(float(got.sum()) - 1.0) < 1e-5 and float((got - want[b]).abs().max()) < 2e-3 * float(want[b].max())
        top2 = torch.topk(ref_logits[b], 2).values
        if float(top2[0] - top2[1]) > 1e-2:
            assert int(tokens[b]) == ids[int(ref_logits[b].argmax())] and max(probs[b], key=probs[b].get) == tok.all_language_codes[int(ref_logits[b].argmax())]
    t1, p1 = model.detect_language(mel[0])
    assert t1.dim() == 0 and int(t1) == int(tokens[0]) and isinstance(p1, dict) and abs(p1["en"] - probs[0]["en"]) < 1e-6
    english_only = model.get_tokenizer()
    english_only.language = None
    with pytest.raises(ValueError, match="language tokens"):
        model.detect_language(mel, english_only)
