#!/usr/bin/env python
"""Runs the reference's OWN Kokoro / KittenTTS source files (``/root/reference/mlx_audio/...``, imported from where they lie, unmodified) on
synthetic checkpoints and stores what they compute under ``tests/golden/ref_*.npz``.  ``tests/test_reference_fixtures_cpu.py`` pins the oracle
(``oracle/kokoro_ref.py``, ``oracle/kitten_ref.py``) to these files, so the oracle is no longer a restatement checked only against itself.

MLX is not installable here (no network; ``uv.lock`` pins mlx 0.31.2): the array library underneath the reference's code is the numpy stand-in
``tests/golden/mlx_shim.py`` (read its header for what that does and does not prove).  Before anything is written, the stand-in itself is
checked against every known-answer vector the reference's tests hold for this path (``tts/tests/test_istftnet_fidelity.py``,
``tts/tests/test_interpolate.py``, ``tts/tests/test_sinegen_length_alignment.py``) by running those assertions through it.

Only runs in the build container (needs /root/reference): ``python tests/golden/make_reference_fixtures.py``.  The synthetic checkpoints are
regenerated from their seeds by the test, so the fixtures hold inputs' seeds + the reference's outputs only.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/mlx_audio"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import mlx_shim  # noqa: E402

mx, nn = mlx_shim.install()


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def import_reference():
    """The reference's modules under their real names, without executing the package ``__init__`` chains (those pull in the HF hub, the CLI ...)."""
    _pkg("mlx_audio", REF)
    _pkg("mlx_audio.tts", f"{REF}/tts")
    _pkg("mlx_audio.tts.models", f"{REF}/tts/models")
    dsp = _load("mlx_audio.dsp", f"{REF}/dsp.py")
    utils = types.ModuleType("mlx_audio.utils")  # mlx_audio/utils.py:31-40 re-exports these two from dsp
    utils.stft, utils.istft = dsp.stft, dsp.istft
    sys.modules["mlx_audio.utils"] = utils
    _load("mlx_audio.tts.models.base", f"{REF}/tts/models/base.py")
    interp = _load("mlx_audio.tts.models.interpolate", f"{REF}/tts/models/interpolate.py")
    _pkg("mlx_audio.tts.models.kokoro", f"{REF}/tts/models/kokoro")
    pipe = types.ModuleType("mlx_audio.tts.models.kokoro.pipeline")  # G2P front end (misaki): not on this path
    pipe.KokoroPipeline = type("KokoroPipeline", (), {})
    sys.modules["mlx_audio.tts.models.kokoro.pipeline"] = pipe
    k_ist = _load("mlx_audio.tts.models.kokoro.istftnet", f"{REF}/tts/models/kokoro/istftnet.py")
    _load("mlx_audio.tts.models.kokoro.modules", f"{REF}/tts/models/kokoro/modules.py")
    kokoro = _load("mlx_audio.tts.models.kokoro.kokoro", f"{REF}/tts/models/kokoro/kokoro.py")
    _pkg("mlx_audio.tts.models.kitten_tts", f"{REF}/tts/models/kitten_tts")
    _load("mlx_audio.tts.models.kitten_tts.quant", f"{REF}/tts/models/kitten_tts/quant.py")
    t_ist = _load("mlx_audio.tts.models.kitten_tts.istftnet", f"{REF}/tts/models/kitten_tts/istftnet.py")
    _load("mlx_audio.tts.models.kitten_tts.modules", f"{REF}/tts/models/kitten_tts/modules.py")
    _load("mlx_audio.tts.models.kitten_tts.preprocess", f"{REF}/tts/models/kitten_tts/preprocess.py")
    kitten = _load("mlx_audio.tts.models.kitten_tts.kitten_tts", f"{REF}/tts/models/kitten_tts/kitten_tts.py")
    return dict(dsp=dsp, interp=interp, k_ist=k_ist, t_ist=t_ist, kokoro=kokoro, kitten=kitten)


def check_shim_against_reference_vectors(R):
    """The reference's own known-answer assertions for this path, run through the stand-in (same numbers as the test files cited)."""
    for ist in (R["k_ist"], R["t_ist"]):
        # tts/tests/test_istftnet_fidelity.py:18-31
        conv = ist.ConvWeighted(1, 1, kernel_size=3, stride=2, padding=0)
        conv.weight_v = mx.array([1.0, 2.0, 3.0]).reshape(1, 3, 1)
        conv.weight_g = mx.array([float(np.sqrt(14.0))]).reshape(1, 1, 1)
        out = conv(mx.array([1.0, 2.0, 3.0, 4.0]).reshape(1, 4, 1), mx.conv_transpose1d)[:, 1:, :]
        np.testing.assert_allclose(np.array(out).reshape(-1), [2.0, 5.0, 4.0, 9.0, 6.0, 13.0, 8.0, 12.0], rtol=1e-4)
        # :34-47 MLXSTFT round trip at unity gain
        stft = ist.MLXSTFT(filter_length=20, hop_length=5, win_length=20)
        t = np.arange(2000, dtype=np.float32)
        x = (0.5 * np.sin(2 * np.pi * 220 * t / 24000)).astype(np.float32)
        rec = np.array(stft.inverse(*stft.transform(mx.array(x)[None, :]))).reshape(-1)[: x.shape[0]]
        np.testing.assert_allclose(rec[20:-20], x[20:-20], atol=1e-3)
    # tts/tests/test_interpolate.py:40-97 and tts/tests/test_sinegen_length_alignment.py:8-17 (vectors transcribed in reference_vectors.json)
    import json

    vec = json.load(open(os.path.join(HERE, "reference_vectors.json")))
    iv = vec["interpolate"]
    f = R["interp"].interpolate
    x = mx.array(np.asarray(iv["nearest_in"], dtype=np.float32).reshape(1, 1, -1))
    np.testing.assert_allclose(np.array(f(x, size=8, mode="nearest")).reshape(-1), iv["nearest_up8"], rtol=iv["rtol"])
    np.testing.assert_allclose(np.array(f(x, size=2, mode="nearest")).reshape(-1), iv["nearest_down2"], rtol=iv["rtol"])
    x = mx.array(np.asarray(iv["linear_in"], dtype=np.float32).reshape(1, 1, -1))
    np.testing.assert_allclose(np.array(f(x, size=7, mode="linear", align_corners=True)).reshape(-1), iv["linear_ac_true_7"], rtol=iv["rtol"])
    np.testing.assert_allclose(np.array(f(x, size=7, mode="linear", align_corners=False)).reshape(-1), iv["linear_ac_false_7"], rtol=iv["rtol"])
    sg = vec["sinegen_shapes"]
    for ist in (R["k_ist"], R["t_ist"]):
        gen = ist.SineGen(24000, upsample_scale=sg["upsample_scale"], harmonic_num=sg["harmonic_num"])
        sine, uv, _ = gen(mx.full((1, sg["length"], 1), sg["f0"]))
        assert list(sine.shape) == sg["sine_shape"] and list(uv.shape) == [1, sg["length"], 1]
    return 4


class Recorder:
    """Wraps bound methods of reference modules to keep their inputs / outputs."""

    def __init__(self):
        self.data = {}

    def wrap(self, obj, attr, name, pick=lambda args, out: out):
        fn = getattr(obj, attr)

        def g(*a, **k):
            out = fn(*a, **k)
            self.data[name] = pick(a, out)
            return out

        setattr(obj, attr, g)


def _np(x):
    return np.asarray(x, dtype=np.float32 if np.asarray(x).dtype.kind == "f" else None)


def run_kokoro(R, seed_w, n_phon, seed_ids, speed, seed_rng, off_grid=False):
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    cfg = S.tiny_config()
    w = S.make_kokoro_weights(cfg, seed=seed_w)
    if off_grid:  # a genuinely float32 checkpoint: no value is bf16-representable
        w = S.as_float32_checkpoint(w, seed=seed_w)
    K = R["kokoro"]
    model = K.Model(K.ModelConfig.from_dict(cfg), repo_id="none")
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    # every parameter the reference's modules own is in the synthetic checkpoint and vice versa (bert.pooler is constructed by CustomAlbert)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    model.eval()  # the reference's loader does (utils.py: base_load_model)
    ids = S.make_phoneme_ids(n_phon, seed=seed_ids)
    inv = {v: k for k, v in cfg["vocab"].items()}
    phonemes = "".join(inv[int(i)] for i in ids[1:-1])
    ref_s = S.make_voice_pack()[len(ids) - 3].numpy()
    rec = Recorder()
    rec.wrap(model.predictor, "F0Ntrain", "f0n", lambda a, out: (np.array(a[0]), np.array(out[0]), np.array(out[1])))
    dec = model.decoder
    rec.wrap(dec, "__call__", "dec_in", lambda a, out: tuple(np.array(v) for v in a))
    # Decoder is invoked as ``decoder(asr, F0, N, s)`` through type(obj).__call__, so patch the class-level entry instead
    cls_call = type(dec).__call__

    def dec_call(self, asr, f0, n, s):
        rec.data["dec_in"] = (np.array(asr), np.array(f0), np.array(n))
        return cls_call(self, asr, f0, n, s)

    type(dec).__call__ = dec_call
    gen = dec.generator
    gcall = type(gen).__call__

    def gen_call(self, x, s, f0):
        rec.data["xg"] = np.array(x)
        return gcall(self, x, s, f0)

    type(gen).__call__ = gen_call
    src = gen.m_source
    scall = type(src).__call__

    def src_call(self, x):
        out = scall(self, x)
        rec.data["har_src"] = np.array(out[0])
        return out

    type(src).__call__ = src_call
    lstm_cls = type(model.predictor.lstm)
    lcall = lstm_cls.__call__
    dur_in = {}

    def lstm_call(self, x, *a, **k):
        if self is model.predictor.lstm:
            dur_in["d"] = np.array(x)
        return lcall(self, x, *a, **k)

    lstm_cls.__call__ = lstm_call
    mx.random.seed(seed_rng)
    try:
        out = model(phonemes, mx.array(ref_s), speed=speed, return_output=True)
    finally:
        type(dec).__call__, type(gen).__call__, type(src).__call__, lstm_cls.__call__ = cls_call, gcall, scall, lcall
    draws = list(mx.random.log)
    uni = [v for kind, shp, v in draws if kind == "uniform"]
    nor = [v for kind, shp, v in draws if kind == "normal"]
    # SineGen: rand_ini [1, 9], noise [1, L, 9]; the third draw [1, L, 1] is SourceModuleHnNSF's unused noise branch (istftnet.py:707-708)
    assert len(uni) == 1 and nor[0].shape[-1] == 9 and len(nor) == 2, [(k, s) for k, s, _ in draws]
    asr, f0, n = rec.data["dec_in"]
    return dict(cfg_kind="kokoro_tiny", seed_w=seed_w, n_phon=n_phon, seed_ids=seed_ids, speed=np.float32(speed),
                pred_dur=np.asarray(out.pred_dur, dtype=np.int32), d=dur_in["d"].astype(np.float32), f0=f0.astype(np.float32), n=n.astype(np.float32),
                asr=asr.astype(np.float32), xg_every8=rec.data["xg"][:, ::8].astype(np.float32), har_src=rec.data["har_src"].astype(np.float32),
                seed_rng=seed_rng, rand_ini=uni[0].astype(np.float32), noise_head=nor[0][0, :4].astype(np.float32),  # the draws are re-made from seed_rng
                audio=np.asarray(out.audio, dtype=np.float32))


def run_kitten(R, seed_w, n_phon, seed_ids, speed, seed_rng, quant):
    from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    cfg = KS.tiny_config()
    w = KS.make_kitten_weights(cfg, seed=seed_w)
    qm = KS.converter_quant_modules(w) if quant else None
    cfg = dict(cfg, activation_quant_modules=qm)
    T = R["kitten"]
    model = T.Model(T.ModelConfig.from_dict(cfg))
    model.load_weights([(k, v.numpy()) for k, v in model.sanitize({k: v for k, v in w.items()}).items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    model.eval()
    flagged = sorted(name for name, m in model.named_modules() if getattr(m, "activation_quant", False))
    all_modules = sorted(name for name, m in model.named_modules() if name)
    ids = S.make_phoneme_ids(n_phon, seed=seed_ids)
    ref_s = S.make_voice_pack()[len(ids) - 3].numpy()
    rec = {}
    dec = model.decoder
    dcall = type(dec).__call__

    def dec_call(self, asr, f0, n, s):
        rec["dec_in"] = (np.array(asr), np.array(f0), np.array(n))
        return dcall(self, asr, f0, n, s)

    type(dec).__call__ = dec_call
    gen = dec.generator
    gcall = type(gen).__call__

    def gen_call(self, x, s, f0):
        rec["xg"] = np.array(x)
        return gcall(self, x, s, f0)

    type(gen).__call__ = gen_call
    lstm_cls = type(model.predictor.lstm)
    lcall = lstm_cls.__call__

    def lstm_call(self, x, *a, **k):
        if self is model.predictor.lstm:
            rec["d"] = np.array(x)
        return lcall(self, x, *a, **k)

    lstm_cls.__call__ = lstm_call
    mx.random.seed(seed_rng)
    try:
        out = model(mx.array(ids.numpy()[None, :], dtype=mx.int32), mx.array(ref_s), speed=speed, return_output=True)
    finally:
        type(dec).__call__, type(gen).__call__, lstm_cls.__call__ = dcall, gcall, lcall
    draws = list(mx.random.log)
    uni = [v for kind, shp, v in draws if kind == "uniform"]
    nor = [v for kind, shp, v in draws if kind == "normal"]
    assert len(uni) == 1 and nor[0].shape[-1] == 9 and len(nor) == 2, [(k, s) for k, s, _ in draws]
    asr, f0, n = rec["dec_in"]
    return dict(cfg_kind="kitten_tiny", seed_w=seed_w, n_phon=n_phon, seed_ids=seed_ids, speed=np.float32(speed), quant=np.int32(bool(quant)),
                flagged_modules=np.array(flagged), all_modules=np.array(all_modules), pred_dur=np.asarray(out.pred_dur, dtype=np.int32), d=rec["d"].astype(np.float32),
                f0=f0.astype(np.float32), n=n.astype(np.float32), asr=asr.astype(np.float32), xg=rec["xg"].astype(np.float32),
                seed_rng=seed_rng, rand_ini=uni[0].astype(np.float32), noise_head=nor[0][0, :4].astype(np.float32),
                audio=np.asarray(out.audio, dtype=np.float32))


class FakeWhisperTokenizer:
    """What DecodingTask reads from a tokenizer (decoding.py:455-510), with the multilingual vocabulary's special ids (oracle.whisper_ref.TokenizerSpec)."""
    eot, sot, translate, transcribe, sot_lm, sot_prev, no_speech, no_timestamps, timestamp_begin = 50257, 50258, 50358, 50359, 50360, 50361, 50362, 50363, 50364
    language_token = 50259
    non_speech_tokens = (1, 2, 7, 8, 9, 10, 14, 25)
    language = "en"

    @property
    def sot_sequence(self):
        return (self.sot, self.language_token, self.transcribe)

    @property
    def sot_sequence_including_notimestamps(self):
        return self.sot_sequence + (self.no_timestamps,)

    def encode(self, text):
        assert text == " "
        return [220]

    def decode(self, tokens):
        return " ".join(str(t) for t in tokens)


def import_whisper():
    _pkg("mlx_audio.stt", f"{REF}/stt")
    _pkg("mlx_audio.stt.models", f"{REF}/stt/models")
    _pkg("mlx_audio.stt.models.whisper", f"{REF}/stt/models/whisper")
    su = types.ModuleType("mlx_audio.stt.utils")
    su.load_audio = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no audio files here"))
    sys.modules["mlx_audio.stt.utils"] = su
    dsp = sys.modules["mlx_audio.dsp"]
    u = sys.modules["mlx_audio.utils"]
    u.hanning, u.mel_filters = dsp.hanning, dsp.mel_filters
    base = "mlx_audio.stt.models.whisper"
    _load(f"{base}.audio", f"{REF}/stt/models/whisper/audio.py")
    _load(f"{base}.tokenizer", f"{REF}/stt/models/whisper/tokenizer.py")
    dec = _load(f"{base}.decoding", f"{REF}/stt/models/whisper/decoding.py")
    _load(f"{base}.timing", f"{REF}/stt/models/whisper/timing.py")
    wh = _load(f"{base}.whisper", f"{REF}/stt/models/whisper/whisper.py")
    return wh, dec


def run_whisper(seed_w, seed_mel, sample_len):
    """The reference's Whisper ``Model`` (encoder, decoder with its KV cache) and ``DecodingTask`` (greedy, all three logit filters) on a tiny
    synthetic checkpoint, float32 (the composition is what is pinned; the fp16 roundings of the released checkpoints are the oracle's own model)."""
    from mlx_audio_amd.stt.models.whisper import synthetic as WS

    wh, dec = import_whisper()
    dims = WS.tiny_dims()
    w = WS.make_whisper_weights(dims, seed=seed_w)
    rd = wh.ModelDimensions(**{k: getattr(dims, k) for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab",
                                                               "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")})
    model = wh.Model(rd, dtype=mx.float32)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected and not mism, (missing[:5], unexpected[:5], mism[:5])
    model.eval()
    model.get_tokenizer = lambda language=None, task="transcribe": FakeWhisperTokenizer()
    mel = WS.make_mel(2, seed=seed_mel, n_frames=2 * dims.n_audio_ctx).numpy()
    xa = model.encoder(mx.array(mel))
    tok = FakeWhisperTokenizer()
    # teacher-forced decoder logits: full context, then one cached step
    ctx = np.array([list(tok.sot_sequence) + [50364, 400, 1200, 31]] * 2, dtype=np.int32)
    logits_full, kv, _ = model.decoder(mx.array(ctx), xa)
    step_tok = np.array([[805], [17]], dtype=np.int32)
    logits_step, kv, _ = model.decoder(mx.array(step_tok), xa, kv_cache=kv)
    out = {}
    for name, kw in (("ts", dict()), ("nots", dict(without_timestamps=True))):
        opts = dec.DecodingOptions(language="en", fp16=False, temperature=0.0, sample_len=sample_len, suppress_tokens="-1", **kw)
        res = dec.decode(model, mx.array(mel), opts)
        ntok = max(len(r.tokens) for r in res)
        toks = np.full((2, ntok), -1, dtype=np.int32)
        for i, r in enumerate(res):
            toks[i, : len(r.tokens)] = r.tokens
        out[f"{name}_tokens"] = toks
        out[f"{name}_avg_logprob"] = np.array([r.avg_logprob for r in res], dtype=np.float64)
        out[f"{name}_no_speech"] = np.array([r.no_speech_prob for r in res], dtype=np.float64)
    # the log-mel front end on 1.5 s of noise + tones (stt/models/whisper/audio.py:41-82 over dsp.stft / mel_filters)
    ga = np.random.default_rng(seed_mel)
    t = np.arange(24000) / 16000.0
    wave = (0.1 * ga.standard_normal(24000) + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t)).astype(np.float32)
    logmel = np.asarray(sys.modules["mlx_audio.stt.models.whisper.audio"].log_mel_spectrogram(wave, n_mels=80, padding=8000))
    return dict(seed_w=seed_w, seed_mel=seed_mel, sample_len=sample_len, non_speech_tokens=np.array(tok.non_speech_tokens, dtype=np.int32),
                logmel=logmel.astype(np.float32),
                xa_every4=np.asarray(xa)[:, :, ::4].astype(np.float32), ctx=ctx, step_tok=step_tok,
                logits_full_last=np.asarray(logits_full)[:, -1, ::16].astype(np.float32), logits_step=np.asarray(logits_step)[:, -1, ::16].astype(np.float32),
                logits_full_argmax=np.asarray(logits_full).argmax(-1).astype(np.int32), **out)


def import_lm_and_mimi():
    """lm/models/{base,cache,...} and codec/models/mimi/** under their real names (package __init__ chains skipped as above)."""
    if "mlx_audio.codec.models.mimi.mimi" in sys.modules:
        return sys.modules["mlx_audio.codec.models.mimi.mimi"]
    _pkg("mlx_audio.lm", f"{REF}/lm")
    _pkg("mlx_audio.lm.models", f"{REF}/lm/models")
    for m in ("base", "cache", "activations", "rope_utils"):
        _load(f"mlx_audio.lm.models.{m}", f"{REF}/lm/models/{m}.py")
    _pkg("mlx_audio.codec", f"{REF}/codec")
    _pkg("mlx_audio.codec.models", f"{REF}/codec/models")
    _pkg("mlx_audio.codec.models.mimi", f"{REF}/codec/models/mimi")
    mods = _pkg("mlx_audio.codec.models.mimi.modules", f"{REF}/codec/models/mimi/modules")
    base = "mlx_audio.codec.models.mimi.modules"
    for m in ("conv", "quantization", "seanet", "transformer"):
        mod = _load(f"{base}.{m}", f"{REF}/codec/models/mimi/modules/{m}.py")
        for k, v in vars(mod).items():  # modules/__init__.py re-exports the public classes
            if not k.startswith("_") and isinstance(v, type):
                setattr(mods, k, v)
    return _load("mlx_audio.codec.models.mimi.mimi", f"{REF}/codec/models/mimi/mimi.py")


def run_mimi(seed_w, seed_codes, n_frames):
    """The reference's ``Mimi.decode`` (quantizer.decode -> ConvTrUpsample1d -> ProjectedTransformer with its KV cache -> SeanetDecoder,
    mimi.py:155-161) on a tiny synthetic checkpoint: one whole-utterance call, and the same codes frame by frame through ``decode_step``."""
    from mlx_audio_amd.codec.models.mimi import mimi as M

    rm = import_lm_and_mimi()
    mods = sys.modules["mlx_audio.codec.models.mimi.modules"]
    c = M.tiny_mimi_config()
    w = M.make_mimi_decoder_weights(c, seed=seed_w)
    ref_cfg = rm.mimi_202407(c.quantizer_nq)
    sc, tc = ref_cfg.seanet, ref_cfg.transformer
    sc.dimension, sc.nfilters, sc.ratios, sc.ksize, sc.residual_ksize, sc.last_ksize, sc.compress = (c.dimension, c.nfilters, list(c.ratios), c.ksize,
                                                                                                    c.residual_ksize, c.last_ksize, c.compress)
    tc.d_model, tc.num_heads, tc.num_layers, tc.dim_feedforward, tc.context, tc.max_seq_len = (c.dimension, c.num_heads, c.num_layers, c.dim_feedforward,
                                                                                                c.context, c.max_seq_len)
    ref_cfg.quantizer_bins, ref_cfg.quantizer_dim = c.quantizer_bins, c.quantizer_dim
    model = rm.Mimi(ref_cfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    # decode-side parameters only: the encoder half, the quantizer's input projections and its ``initialized`` flag (a training buffer) are not on the path
    dec_missing = [m for m in missing if not m.startswith(("encoder", "downsample", "quantizer.rvq_first.input_proj", "quantizer.rvq_rest.input_proj"))
                   and not m.endswith(".codebook.initialized")]
    assert not unexpected and not dec_missing and not mism, (dec_missing[:8], unexpected[:8], mism[:4])
    for _, m in model.named_modules():  # what load_pytorch_weights does after load_weights (mimi.py:252-260): derived tensors of codebooks / conv-transposes
        if isinstance(m, (mods.EuclideanCodebook, sys.modules["mlx_audio.codec.models.mimi.modules.conv"].ConvTranspose1d)):
            m.update_in_place()
    model.eval()
    codes = M.make_codes(2, n_frames, c, seed=seed_codes).numpy()
    pcm = np.asarray(model.decode(mx.array(codes.astype(np.int32))))
    model.reset_state()
    steps = [np.asarray(model.decode_step(mx.array(codes[:, :, t:t + 1].astype(np.int32)))) for t in range(n_frames)]
    pcm_steps = np.concatenate(steps, axis=-1)
    return dict(seed_w=seed_w, seed_codes=seed_codes, n_frames=n_frames, pcm=pcm.astype(np.float32), pcm_steps=pcm_steps.astype(np.float32))


def run_mimi_encode(seed_w, seed_audio, n_samples):
    """The reference's ``Mimi.encode`` (SeanetEncoder -> encoder_transformer with its KV cache -> ConvDownsample1d -> SplitResidualVectorQuantizer.encode,
    mimi.py:146-153) on a tiny synthetic checkpoint and a seeded clip whose length is NOT a multiple of the frame size (the extra right padding of
    every strided conv, conv.py:163-173, is on the path); also the latent in front of the quantiser and the decode of the codes (round trip)."""
    from mlx_audio_amd.codec.models.mimi import mimi as M

    rm = import_lm_and_mimi()
    mods = sys.modules["mlx_audio.codec.models.mimi.modules"]
    c = M.tiny_mimi_config()
    w = {**M.make_mimi_decoder_weights(c, seed=seed_w), **M.make_mimi_encoder_weights(c, seed=seed_w)}
    ref_cfg = rm.mimi_202407(c.quantizer_nq)
    sc, tc = ref_cfg.seanet, ref_cfg.transformer
    sc.dimension, sc.nfilters, sc.ratios, sc.ksize, sc.residual_ksize, sc.last_ksize, sc.compress = (c.dimension, c.nfilters, list(c.ratios), c.ksize,
                                                                                                    c.residual_ksize, c.last_ksize, c.compress)
    tc.d_model, tc.num_heads, tc.num_layers, tc.dim_feedforward, tc.context, tc.max_seq_len = (c.dimension, c.num_heads, c.num_layers, c.dim_feedforward,
                                                                                                c.context, c.max_seq_len)
    ref_cfg.quantizer_bins, ref_cfg.quantizer_dim = c.quantizer_bins, c.quantizer_dim
    model = rm.Mimi(ref_cfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    missing = [m for m in missing if not m.endswith(".codebook.initialized")]   # a training buffer
    assert not unexpected and not missing and not mism, (missing[:8], unexpected[:8], mism[:4])
    for _, m in model.named_modules():
        if isinstance(m, (mods.EuclideanCodebook, sys.modules["mlx_audio.codec.models.mimi.modules.conv"].ConvTranspose1d)):
            m.update_in_place()
    model.eval()
    pcm = M.make_pcm(2, n_samples, seed=seed_audio).numpy()
    codes = np.asarray(model.encode(mx.array(pcm)))
    # the same three stages by hand, for the latent (encode() resets the state itself)
    model.encoder.reset_state()
    for cch in model.encoder_cache:
        cch.reset() if hasattr(cch, "reset") else None
    model.reset_state()
    xs = model.encoder(mx.array(pcm))
    xs = model.encoder_transformer(xs, cache=model.encoder_cache)[0]
    z = np.asarray(model.downsample(xs))
    back = np.asarray(model.decode(mx.array(codes.astype(np.int32))))
    return dict(seed_w=seed_w, seed_audio=seed_audio, n_samples=n_samples, codes=codes.astype(np.int64), latent=z.astype(np.float32), decoded=back.astype(np.float32))


def run_qwen3_tokenizer_encode(seed_w, seed_audio, n_samples):
    """The reference's ``Qwen3TTSSpeechTokenizerEncoder.encode`` (speech_tokenizer.py:957-1058: the Mimi modules under the tokenizer's configuration --
    non-traditional RoPE, an explicit causal mask, ``codes[:, :valid_num_quantizers]``) on a tiny synthetic checkpoint handed over in the HuggingFace
    form and passed through the reference's own ``sanitize``."""
    from mlx_audio_amd.codec.models.mimi import mimi as M

    if "mlx_audio.tts.models.qwen3_tts.qwen3_tts" not in sys.modules:
        run_sampler(0)  # loads the qwen3_tts modules
    st = sys.modules["mlx_audio.tts.models.qwen3_tts.speech_tokenizer"]
    cfgm = sys.modules["mlx_audio.tts.models.qwen3_tts.config"]
    import pt_layouts as PT

    c = M.tiny_mimi_config()
    mw = {**M.make_mimi_decoder_weights(c, seed=seed_w), **M.make_mimi_encoder_weights(c, seed=seed_w)}
    ec = cfgm.Qwen3TTSTokenizerEncoderConfig(hidden_size=c.dimension, num_filters=c.nfilters, upsampling_ratios=list(c.ratios), kernel_size=c.ksize,
                                             residual_kernel_size=c.residual_ksize, last_kernel_size=c.last_ksize, compress=c.compress,
                                             num_attention_heads=c.num_heads, num_key_value_heads=c.num_heads, head_dim=c.dimension // c.num_heads,
                                             num_hidden_layers=c.num_layers, intermediate_size=c.dim_feedforward, sliding_window=c.context,
                                             max_position_embeddings=c.max_seq_len, num_quantizers=c.quantizer_nq, codebook_size=c.quantizer_bins,
                                             codebook_dim=c.quantizer_dim, vector_quantization_hidden_dimension=c.quantizer_dim)
    enc = st.Qwen3TTSSpeechTokenizerEncoder(ec)
    san = st.Qwen3TTSSpeechTokenizer.sanitize({k: mx.array(v.numpy()) for k, v in PT.qwen3_tokenizer_encoder_checkpoint(mw, c.num_layers, c.quantizer_nq).items()})
    enc.load_weights([(k[len("encoder_model."):], v) for k, v in san.items() if k.startswith("encoder_model.")])
    missing, unexpected, mism = enc._load_report
    missing = [m for m in missing if not m.endswith(".codebook.initialized")]
    assert not unexpected and not missing and not mism, (missing[:8], unexpected[:8], mism[:4])
    mods = sys.modules["mlx_audio.codec.models.mimi.modules"]
    for _, m in enc.named_modules():
        if isinstance(m, mods.EuclideanCodebook):
            m.update_in_place()
    pcm = M.make_pcm(2, n_samples, seed=seed_audio).numpy()
    codes = np.asarray(enc.encode(mx.array(pcm)))
    return dict(seed_w=seed_w, seed_audio=seed_audio, n_samples=n_samples, codes=codes.astype(np.int64), valid=int(enc.valid_num_quantizers))


def ref_qwen3_talker(seed_w):
    """The reference's ``Qwen3TTSTalkerForConditionalGeneration`` on the tiny synthetic checkpoint of this package's generator: (model, cfg, reference cfg)."""
    from dataclasses import asdict

    from mlx_audio_amd.tts.models.qwen3_tts import talker as T

    import_lm_and_mimi()
    if "mlx_audio.tts.models.qwen3_tts.config" not in sys.modules:
        _pkg("mlx_audio.tts.models.qwen3_tts", f"{REF}/tts/models/qwen3_tts")
        _load("mlx_audio.tts.models.qwen3_tts.config", f"{REF}/tts/models/qwen3_tts/config.py")
    rc = sys.modules["mlx_audio.tts.models.qwen3_tts.config"]
    if "mlx_audio.tts.models.qwen3_tts.talker" not in sys.modules:
        _load("mlx_audio.tts.models.qwen3_tts.talker", f"{REF}/tts/models/qwen3_tts/talker.py")
    rt = sys.modules["mlx_audio.tts.models.qwen3_tts.talker"]
    cfg = T.tiny_talker_config()
    w = T.make_talker_weights(cfg, seed=seed_w)
    d = asdict(cfg)
    cpd = d.pop("code_predictor_config")
    known = set(rc.Qwen3TTSTalkerConfig.__dataclass_fields__)
    cp_known = set(rc.Qwen3TTSTalkerCodePredictorConfig.__dataclass_fields__)
    rcfg = rc.Qwen3TTSTalkerConfig(code_predictor_config=rc.Qwen3TTSTalkerCodePredictorConfig(**{k: v for k, v in cpd.items() if k in cp_known}),
                                   **{k: v for k, v in d.items() if k in known})
    model = rt.Qwen3TTSTalkerForConditionalGeneration(rcfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected and not mism, (missing[:8], unexpected[:8], mism[:4])
    model.eval()
    return model, cfg, rcfg


def run_qwen3_generate_loop(seed_w):
    """The reference's single-utterance ``Model.generate`` loop (qwen3_tts.py:1268-1420: prefill, suppressed + repetition-penalised arg-max of the first
    code, code predictor, next input = trailing text or tts_pad + the sum of the 16 code embeddings, EOS test before the frame is kept) on the tiny
    reference talker, greedy, with a character tokenizer and the speech tokenizer replaced by a recorder -- and ``_next_batch_input_embeds``
    (:993-1015), the batched loop's trailing-text rule, in both of its modes.  Two runs: a frame budget of 7, then the same text with the EOS id set to
    the first code the budget run chose at frame 4 (so the EOS branch is taken there)."""
    import pt_layouts as PT

    q = import_qwen3_model()
    model, cfg, rcfg = ref_qwen3_talker(seed_w)
    tok = PT.QwenCharTokenizer(vocab=cfg.text_vocab_size)
    seen = {}
    for name, value in PT.QWEN3_LOOP_CODEC_IDS.items():   # the special codec ids of the real checkpoint lie beyond the tiny vocabulary
        setattr(rcfg, name, value)

    class Host:
        generate = q.Model.generate
        _prepare_generation_inputs = q.Model._prepare_generation_inputs
        _sample_token = q.Model._sample_token
        _next_batch_input_embeds = q.Model._next_batch_input_embeds
        _codec_embeds_for_tokens = q.Model._codec_embeds_for_tokens
        tokenizer = tok
        talker = model
        speaker_encoder = None
        sample_rate = 24000
        supported_speakers = []
        config = types.SimpleNamespace(talker_config=rcfg, tts_model_type="base", tts_bos_token_id=cfg.text_vocab_size - 3,
                                       tts_eos_token_id=cfg.text_vocab_size - 2, tts_pad_token_id=cfg.text_vocab_size - 1)
        speech_tokenizer = types.SimpleNamespace(has_encoder=False, decode=lambda codes: (seen.__setitem__("codes", np.asarray(codes)[0].astype(np.int32)),
                                                                                           (mx.zeros((1, codes.shape[1] * 8)), mx.array([codes.shape[1] * 8])))[1])

    host = Host()
    text = "Hi, how are you?"
    logits_seen = []
    orig_sample = Host._sample_token

    def spy(self, logits, **kw):
        if kw.get("suppress_tokens"):
            logits_seen.append(np.asarray(logits)[0, -1].astype(np.float32))
        return orig_sample(self, logits, **kw)

    Host._sample_token = spy
    ex, etr, epad = host._prepare_generation_inputs(text, language="auto", speaker=None)
    res = list(host.generate(text, temperature=0.0, max_tokens=7, lang_code="auto"))
    budget_codes, budget_logits = seen["codes"].copy(), np.stack(logits_seen)
    assert len(res) == 1 and budget_codes.shape == (7, cfg.num_code_groups) and res[0].token_count == 7
    eos_saved = rcfg.codec_eos_token_id
    eos_new = int(budget_codes[4, 0])
    assert eos_new not in budget_codes[:4, 0].tolist()
    rcfg.codec_eos_token_id = eos_new
    logits_seen.clear()
    try:
        res2 = list(host.generate(text, temperature=0.0, max_tokens=7, lang_code="auto"))
    finally:
        rcfg.codec_eos_token_id = eos_saved
    eos_codes = seen["codes"].copy()
    assert res2[0].token_count == 4 and np.array_equal(eos_codes, budget_codes[:4])
    # a prompt whose trailing text (2 tokens + tts_eos) runs out after three frames: the tts_pad branch of the loop
    Host.tokenizer = types.SimpleNamespace(encode=lambda t: [11 + (7 * i + len(t)) % 300 for i in range(11)])
    logits_seen.clear()
    sx, strl, spad = host._prepare_generation_inputs(text, language="auto", speaker=None)
    res3 = list(host.generate(text, temperature=0.0, max_tokens=7, lang_code="auto"))
    short_codes, short_logits = seen["codes"].copy(), np.stack(logits_seen)
    assert np.asarray(strl).shape[1] == 3 and short_codes.shape[0] == 7 and len(res3) == 1
    Host.tokenizer = tok
    # the batched loop's trailing-text rule
    g = np.random.default_rng(seed_w + 1)
    H = cfg.hidden_size
    tr = (g.standard_normal((3, 4, H)) * 0.5).astype(np.float32)
    pad = (g.standard_normal((1, 1, H)) * 0.5).astype(np.float32)
    idx = np.array([[1], [3], [6]], dtype=np.int32)
    codes = [mx.array(budget_codes[f:f + 1, i:i + 1].repeat(3, 0)) for f, i in ((0, 0),)] + \
            [mx.array(budget_codes[0:1, i:i + 1].repeat(3, 0)) for i in range(1, cfg.num_code_groups)]
    nxt = {str(int(flag)): np.asarray(host._next_batch_input_embeds(mx.array(tr), mx.array(pad), mx.array(idx), codes, pad_when_index_clamped=flag))
           .astype(np.float32) for flag in (False, True)}
    return dict(seed_w=seed_w, text=text, prefill=np.asarray(ex).astype(np.float32), trailing=np.asarray(etr).astype(np.float32), pad=np.asarray(epad).astype(np.float32),
                short_prefill=np.asarray(sx).astype(np.float32), short_trailing=np.asarray(strl).astype(np.float32), short_codes=short_codes,
                short_logits=short_logits, budget_codes=budget_codes, budget_logits=budget_logits, eos_id=eos_new, eos_codes=eos_codes, eos_logits=np.stack(logits_seen),
                next_trailing=tr, next_pad=pad, next_idx=idx, next_codes=budget_codes[0], next_embeds_unclamped=nxt["0"], next_embeds_clamped=nxt["1"])


def run_qwen3_talker(seed_w, seed_in):
    """The reference's ``Qwen3TTSTalkerForConditionalGeneration`` (talker stack with MRoPE position ids, q / k norms, GQA, KV cache, ``codec_head``;
    talker.py:229-500, 767-822) and ``Qwen3TTSTalkerCodePredictor`` (talker.py:503-764) stepped exactly like ``_predict_code_tokens``
    (qwen3_tts.py:941-983) with forced codes, on a tiny synthetic checkpoint: prefill of 11 positions, then two single-position steps."""
    model, cfg, rcfg = ref_qwen3_talker(seed_w)
    g = np.random.default_rng(seed_in)
    B, L, H = 2, 11, cfg.hidden_size
    prefill = (g.standard_normal((B, L, H)) * 0.5).astype(np.float32)
    steps = (g.standard_normal((2, B, 1, H)) * 0.5).astype(np.float32)
    cache = model.make_cache()
    logits0, hid0 = model(mx.array(prefill), cache=cache)
    outs = [(np.asarray(logits0)[:, -1], np.asarray(hid0)[:, -1])]
    for s_ in steps:
        lg, hd = model(mx.array(s_), cache=cache)
        outs.append((np.asarray(lg)[:, -1], np.asarray(hd)[:, -1]))
    # code predictor, teacher-forced on fixed codes, from the hidden state of the last talker step
    ng = cfg.num_code_groups
    forced = g.integers(0, cfg.code_predictor_config.vocab_size, size=(B, ng)).astype(np.int32)
    forced[:, 0] = g.integers(0, cfg.vocab_size - 1024, size=B)
    cp_cache = model.code_predictor.make_cache()
    hidden = mx.array(outs[-1][1][:, None, :])
    cp_logits = []
    for i in range(ng - 1):
        if i == 0:
            e0 = model.get_input_embeddings()(mx.array(forced[:, 0:1]))
            inp = mx.concatenate([hidden, e0], axis=1)
        else:
            inp = model.code_predictor.codec_embedding[i - 1](mx.array(forced[:, i:i + 1]))
        lg, cp_cache, _ = model.code_predictor(inp, cache=cp_cache, generation_step=i)
        cp_logits.append(np.asarray(lg)[:, -1])
    text_ids = g.integers(0, cfg.text_vocab_size, size=(B, 5)).astype(np.int32)
    tproj = np.asarray(model.text_projection(model.get_text_embeddings()(mx.array(text_ids))))
    return dict(seed_w=seed_w, seed_in=seed_in, prefill=prefill, steps=steps, forced=forced, text_ids=text_ids, text_projection=tproj.astype(np.float32),
                logits=np.stack([o[0] for o in outs]).astype(np.float32), hidden=np.stack([o[1] for o in outs]).astype(np.float32),
                cp_logits=np.stack(cp_logits).astype(np.float32))


def run_qwen3_codec(seed_w, seed_codes, n_frames):
    """The reference's ``Qwen3TTSSpeechTokenizerDecoder`` (split RVQ decode, pre_conv, 8-layer-style transformer with LayerScale and sliding window,
    ConvNeXt up-samplers, SnakeBeta decoder blocks: speech_tokenizer.py:32-955), whole-utterance ``__call__`` and ``chunked_decode``."""
    from dataclasses import asdict

    from mlx_audio_amd.tts.models.qwen3_tts import synthetic as QS

    import_lm_and_mimi()
    if "mlx_audio.tts.models.qwen3_tts.config" not in sys.modules:
        _pkg("mlx_audio.tts.models.qwen3_tts", f"{REF}/tts/models/qwen3_tts")
        _load("mlx_audio.tts.models.qwen3_tts.config", f"{REF}/tts/models/qwen3_tts/config.py")
    rc = sys.modules["mlx_audio.tts.models.qwen3_tts.config"]
    st = _load("mlx_audio.tts.models.qwen3_tts.speech_tokenizer", f"{REF}/tts/models/qwen3_tts/speech_tokenizer.py")
    cfg = QS.tiny_codec_config()
    w = QS.make_codec_decoder_weights(cfg, seed=seed_w)
    known = set(rc.Qwen3TTSTokenizerDecoderConfig.__dataclass_fields__)
    rcfg = rc.Qwen3TTSTokenizerDecoderConfig(**{k: v for k, v in asdict(cfg).items() if k in known})
    model = st.Qwen3TTSSpeechTokenizerDecoder(rcfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not unexpected and not mism, (unexpected[:8], mism[:4])
    model.eval()
    for _, m in model.named_modules():
        if hasattr(m, "update_in_place"):
            m.update_in_place()
    codes = QS.make_codes(2, n_frames, cfg, seed=seed_codes).numpy().astype(np.int32)
    audio = np.asarray(model(mx.array(codes)))
    chunked = np.asarray(model.chunked_decode(mx.array(codes), chunk_size=12, left_context_size=5))
    return dict(seed_w=seed_w, seed_codes=seed_codes, n_frames=n_frames, missing=np.array(missing), audio=audio.astype(np.float32),
                chunked=chunked.astype(np.float32))


def import_sesame():
    """The reference's ``tts/models/sesame/sesame.py`` under its real name."""
    name = "mlx_audio.tts.models.sesame.sesame"
    if name in sys.modules:
        return sys.modules[name]
    import_lm_and_mimi()
    _load("mlx_audio.lm.models.llama", f"{REF}/lm/models/llama.py")
    _pkg("mlx_audio.tts.models.sesame", f"{REF}/tts/models/sesame")
    aio = types.ModuleType("mlx_audio.audio_io")
    aio.read = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no audio files here"))
    sys.modules["mlx_audio.audio_io"] = aio
    mimi_pkg = sys.modules["mlx_audio.codec.models.mimi"]
    mimi_pkg.Mimi = sys.modules["mlx_audio.codec.models.mimi.mimi"].Mimi
    mimi_pkg.MimiStreamingDecoder = type("MimiStreamingDecoder", (), {})
    _load("mlx_audio.tts.models.sesame.attention", f"{REF}/tts/models/sesame/attention.py")
    # sesame.py imports the HF tokenizer stack at module level (text tokenisation, not on this path); ``transformers`` probes for a real ``mlx``
    # and trips over the stand-in, so both names are stubbed for the duration of this one import
    u = sys.modules["mlx_audio.utils"]
    u.load_audio = u.resample_audio = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no audio files here"))
    saved = {k: sys.modules.get(k) for k in ("transformers", "tokenizers", "tokenizers.processors")}
    tf = types.ModuleType("transformers")
    tf.AutoTokenizer = None
    tk, tkp = types.ModuleType("tokenizers"), types.ModuleType("tokenizers.processors")
    tkp.TemplateProcessing = None
    tk.processors = tkp
    sys.modules.update({"transformers": tf, "tokenizers": tk, "tokenizers.processors": tkp})
    try:
        rs = _load("mlx_audio.tts.models.sesame.sesame", f"{REF}/tts/models/sesame/sesame.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return rs


def run_csm_generate():
    """The reference's CSM ``Model.generate`` (sesame.py:730-866) with its prompt-frame builders (``_tokenize_text_segment`` / ``_tokenize_audio`` /
    ``_tokenize_segment``, :502-575) on scripted parts: the text tokenizer is one id per character, ``Mimi.encode`` is ``pt_layouts.csm_fake_codes``,
    ``generate_frame`` replays scripted frames and records what it is handed, ``generate_result`` records the frames it is given.  Cases:
    ``pt_layouts.CSM_GENERATE_CASES`` (speaker prefix with / without space, prompt splitting, context with and without voice matching, reference audio,
    streaming chunks, the frame budget, a list of prompts)."""
    import json

    import pt_layouts as PT

    rs = import_sesame()
    K = PT.CSM_CODEBOOKS
    tok = PT.CsmCharTokenizer()
    saved_load = rs.load_audio
    rs.load_audio = lambda a, sample_rate=None, **k: a
    out = []
    try:
        for case in PT.CSM_GENERATE_CASES:
            prompts = []

            class Engine:
                args = types.SimpleNamespace(audio_num_codebooks=K)

                def reset_caches(self):
                    prompts.append(dict(calls=[]))

                def generate_frame(self, tokens, mask, pos, sampler):
                    i, calls = len(prompts) - 1, prompts[-1]["calls"]
                    j = len(calls)
                    calls.append(dict(tokens=np.asarray(tokens)[0].astype(int).tolist(), mask=np.asarray(mask)[0].astype(int).tolist(),
                                      pos=np.asarray(pos)[0].astype(int).tolist()))
                    frame = PT.csm_frame(i, j) if j < case["frames"][i] else [0] * K
                    return mx.array(np.array([frame], dtype=np.int32))

            class Host:
                _tokenize_text_segment = rs.Model._tokenize_text_segment
                _tokenize_audio = rs.Model._tokenize_audio
                _tokenize_segment = rs.Model._tokenize_segment
                generate = rs.Model.generate
                sample_rate = 24000
                tokenizer_repo = None
                _frame_size = K + 1
                _speaker_prefix_space = case["cfg"]["speaker_prefix_space"]
                _default_voice_match = case["cfg"]["voice_match"]
                _use_default_voice_prompt = False
                model = Engine()
                _text_tokenizer = types.SimpleNamespace(encode=lambda text, return_tensors=None: mx.array(np.array([tok.ids(text)], dtype=np.int32)))
                _audio_tokenizer = types.SimpleNamespace(encode=lambda x: mx.array(PT.csm_fake_codes(np.asarray(x)[0, 0]))[None])
                _streaming_decoder = types.SimpleNamespace(reset=lambda: None)

                def generate_result(self, samples, start_time, stream=False):
                    return dict(n=len(samples), stream=bool(stream), frames=[np.asarray(f)[0].astype(int).tolist() for f in samples])

            kw = dict(case["kw"])
            if "context" in kw:
                kw["context"] = [rs.Segment(speaker=sp, text=t, audio=mx.array(PT.csm_audio(*au))) for sp, t, au in kw["context"]]
            if "ref_audio" in kw:
                kw["ref_audio"] = mx.array(PT.csm_audio(*kw["ref_audio"]))
            results = list(Host().generate(case["text"], sampler=object(), **kw))
            out.append(dict(name=case["name"], prompts=prompts, results=results))
    finally:
        rs.load_audio = saved_load
    with open(os.path.join(HERE, "ref_csm_generate.json"), "w") as f:
        json.dump(out, f)
    return [(c["name"], [len(p["calls"]) for p in c["prompts"]], [(r["n"], r["stream"]) for r in c["results"]]) for c in out]


def run_csm(seed_w, seed_in, n_frames):
    """The reference's ``SesameModel`` (sesame.py:301-425: two ``LlamaModel`` stacks of lm/models/llama.py with the ``Llama3ScaledRoPE`` attention of
    sesame/attention.py, summed audio + text embeddings, ``codebook0_head``, depth decoder with ``audio_head``) -- ``generate_frame`` called
    ``n_frames`` times on a prompt, with a sampler that returns forced codes and records the logits it was handed."""
    from mlx_audio_amd.tts.models.sesame import engine as E

    rs = import_sesame()
    cfg = E.tiny_csm()
    w = E.make_csm_weights(cfg, seed=seed_w)

    def llama_fields(sc):
        return dict(num_hidden_layers=sc.n_layers, num_attention_heads=sc.n_heads, num_key_value_heads=sc.n_kv_heads, head_dim=sc.head_dim,
                    hidden_size=sc.d_model, intermediate_size=sc.d_ff, rms_norm_eps=sc.norm_eps, max_position_embeddings=sc.max_pos,
                    attention_bias=False, mlp_bias=False, rope_theta=sc.rope_theta,
                    rope_scaling={"factor": sc.rope_llama3_factor, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                                  "original_max_position_embeddings": 8192, "rope_type": "llama3"})

    b, d_ = cfg.backbone, cfg.decoder
    config = dict(model_type="csm", backbone_flavor="none", decoder_flavor="none", text_vocab_size=cfg.text_vocab_size, audio_vocab_size=cfg.audio_vocab_size,
                  audio_num_codebooks=cfg.audio_num_codebooks, attention_dropout=0.0, audio_eos_token_id=0, audio_token_id=0, bos_token_id=0,
                  codebook_eos_token_id=0, codebook_pad_token_id=0, hidden_act="silu", initializer_range=0.02, num_codebooks=cfg.audio_num_codebooks,
                  pad_token_id=0, tie_codebooks_embeddings=True, tie_word_embeddings=False, use_cache=True, vocab_size=cfg.text_vocab_size,
                  depth_decoder_config=dict(attention_dropout=0.0, backbone_hidden_size=b.d_model, hidden_act="silu", initializer_range=0.02,
                                            model_type="csm_depth_decoder_model", num_codebooks=cfg.audio_num_codebooks, use_cache=True,
                                            vocab_size=cfg.audio_vocab_size, **llama_fields(d_)),
                  **llama_fields(b))
    model = rs.SesameModel(config)
    names = {"wq": "self_attn.q_proj", "wk": "self_attn.k_proj", "wv": "self_attn.v_proj", "wo": "self_attn.o_proj", "w_gate": "mlp.gate_proj",
             "w_up": "mlp.up_proj", "w_down": "mlp.down_proj", "attn_norm": "input_layernorm", "mlp_norm": "post_attention_layernorm"}
    ref_w = {}
    for k, v in w.items():  # canonical stack names (mlx_audio_amd/lm/stack.py) -> the reference's module paths
        parts = k.split(".")
        if parts[0] in ("backbone", "decoder") and parts[1] == "layers":
            ref_w[".".join([parts[0], "layers", parts[2], names[parts[3]], parts[4]])] = v
        elif parts[0] in ("backbone", "decoder") and parts[1] == "final_norm":
            ref_w[f"{parts[0]}.norm.{parts[2]}"] = v
        else:
            ref_w[k] = v
    model.load_weights([(k, v.numpy()) for k, v in ref_w.items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected and not mism, (missing[:8], unexpected[:8], mism[:4])
    model.eval()
    model.setup_caches(2)
    g = np.random.default_rng(seed_in)
    B, S, ncb = 2, 7, cfg.audio_num_codebooks
    toks = np.zeros((B, S, ncb + 1), dtype=np.int32)
    mask = np.zeros((B, S, ncb + 1), dtype=bool)
    toks[:, :4, -1] = g.integers(1, cfg.text_vocab_size, size=(B, 4))      # text rows
    mask[:, :4, -1] = True
    toks[:, 4:, :-1] = g.integers(1, cfg.audio_vocab_size, size=(B, S - 4, ncb))   # audio rows
    mask[:, 4:, :-1] = True
    forced = g.integers(1, cfg.audio_vocab_size, size=(n_frames, B, ncb)).astype(np.int32)
    logits = []
    cur_t, cur_m = toks, mask
    for f in range(n_frames):
        rec = []

        def sampler(lg, rec=rec, f=f):
            rec.append(np.asarray(lg))
            return mx.array(forced[f][:, len(rec) - 1])

        pos = np.broadcast_to(np.arange(cur_t.shape[1])[None, :], (B, cur_t.shape[1]))
        out = np.asarray(model.generate_frame(mx.array(cur_t), mx.array(cur_m.astype(np.float32)), mx.array(pos), sampler))
        assert np.array_equal(out, forced[f])
        logits.append(np.stack(rec))
        cur_t = np.concatenate([out, np.zeros((B, 1), np.int32)], axis=1)[:, None, :]
        cur_m = np.concatenate([np.ones((B, ncb), bool), np.zeros((B, 1), bool)], axis=1)[:, None, :]
    return dict(seed_w=seed_w, seed_in=seed_in, n_frames=n_frames, prompt_tokens=toks, prompt_mask=mask, forced=forced,
                logits=np.stack(logits).astype(np.float32))


def _codec_pkgs():
    if "mlx_audio.codec.models" not in sys.modules:
        _pkg("mlx_audio.codec", f"{REF}/codec")
        _pkg("mlx_audio.codec.models", f"{REF}/codec/models")


def run_dac(seed_w, seed_codes, n_frames):
    """The reference's ``DAC.quantizer.from_codes`` + ``DAC.decode`` (codec/models/descript/{dac.py, nn/layers.py, nn/quantize.py}), small widths, all four rates."""
    from mlx_audio_amd.codec.models.descript import make_dac_weights

    _codec_pkgs()
    base = "mlx_audio.codec.models.descript"
    _pkg(base, f"{REF}/codec/models/descript")
    _pkg(f"{base}.nn", f"{REF}/codec/models/descript/nn")
    _load(f"{base}.base", f"{REF}/codec/models/descript/base.py")
    _load(f"{base}.nn.layers", f"{REF}/codec/models/descript/nn/layers.py")
    _load(f"{base}.nn.quantize", f"{REF}/codec/models/descript/nn/quantize.py")
    rd = _load(f"{base}.dac", f"{REF}/codec/models/descript/dac.py")
    rates, dim, latent, nq, csize, cdim = [8, 5, 4, 2], 64, 32, 3, 128, 8
    w = make_dac_weights(dim, rates, latent, nq, csize, cdim, seed=seed_w)
    model = rd.DAC(encoder_dim=16, encoder_rates=[2, 4, 5, 8], latent_dim=latent, decoder_dim=dim, decoder_rates=rates, n_codebooks=nq,
                   codebook_size=csize, codebook_dim=cdim, sample_rate=16000)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    dec_missing = [m for m in missing if not m.startswith("encoder.") and ".in_proj." not in m and ".rel_pos.inv_freq" not in m]  # inv_freq: a constant the module builds itself
    assert not unexpected and not dec_missing and not mism, (dec_missing[:8], unexpected[:8], mism[:4])
    model.eval()
    codes = np.random.default_rng(seed_codes).integers(0, csize, size=(2, nq, n_frames)).astype(np.int32)
    z, zp, _ = model.quantizer.from_codes(mx.array(codes))
    audio = np.asarray(model.decode(z))
    return dict(seed_w=seed_w, codes=codes, z=np.asarray(z).astype(np.float32), audio=audio.astype(np.float32))


def _load_dac_modules():
    _codec_pkgs()
    base = "mlx_audio.codec.models.descript"
    _pkg(base, f"{REF}/codec/models/descript")
    _pkg(f"{base}.nn", f"{REF}/codec/models/descript/nn")
    _load(f"{base}.base", f"{REF}/codec/models/descript/base.py")
    _load(f"{base}.nn.layers", f"{REF}/codec/models/descript/nn/layers.py")
    _load(f"{base}.nn.quantize", f"{REF}/codec/models/descript/nn/quantize.py")
    return _load(f"{base}.dac", f"{REF}/codec/models/descript/dac.py")


DAC_ENC_TINY = dict(encoder_dim=16, encoder_rates=[2, 4, 5, 8], latent_dim=32, decoder_dim=64, decoder_rates=[8, 5, 4, 2], n_codebooks=4, codebook_size=128,
                    codebook_dim=8, sample_rate=16000)


def run_dac_encode(seed_w, seed_audio, n_samples):
    """The reference's ``DAC.encode`` (dac.py:184-192: Encoder -> ResidualVectorQuantize.__call__, nn/quantize.py:17-127) and ``DAC.__call__`` on a seeded
    checkpoint: even and odd strides, a length that is not a whole number of hops, all codebooks and ``n_quantizers = 2``."""
    from mlx_audio_amd.codec.models.descript import make_dac_encoder_weights, make_dac_weights

    rd = _load_dac_modules()
    c = DAC_ENC_TINY
    w = make_dac_weights(c["decoder_dim"], c["decoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_size"], c["codebook_dim"], seed=seed_w)
    w.update(make_dac_encoder_weights(c["encoder_dim"], c["encoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_dim"], seed=seed_w))
    model = rd.DAC(**c)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not unexpected and not missing and not mism, (missing[:8], unexpected[:8], mism[:4])
    model.eval()
    g = np.random.default_rng(seed_audio)
    t = np.arange(n_samples) / c["sample_rate"]
    audio = np.stack([0.5 * np.sin(2 * np.pi * (180 + 90 * b) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t)) + 0.15 * g.standard_normal(n_samples) for b in range(2)])
    audio = audio[:, None, :].astype(np.float32)
    z, codes, latents, commit, cbl = model.encode(mx.array(audio))
    z2, codes2, latents2, commit2, _ = model.encode(mx.array(audio), 2)
    enc = model.encoder(mx.array(audio).moveaxis(1, 2))
    out = model(mx.array(audio), c["sample_rate"])
    return dict(seed_w=seed_w, config=json.dumps(c), audio=audio, enc=_np(enc), z=_np(z), codes=np.asarray(codes).astype(np.int32), latents=_np(latents),
                commitment_loss=np.float32(np.asarray(commit)), codebook_loss=np.float32(np.asarray(cbl)), z_nq2=_np(z2),
                codes_nq2=np.asarray(codes2).astype(np.int32), latents_nq2=_np(latents2), commitment_loss_nq2=np.float32(np.asarray(commit2)),
                call_audio=_np(out["audio"]), call_codes=np.asarray(out["codes"]).astype(np.int32), call_z=_np(out["z"]))


def run_dac_compress(seed_w, seed_audio):
    """The reference's ``CodecMixin.compress`` / ``decompress`` (codec/models/descript/base.py:123-231) on the seeded checkpoint of ``run_dac_encode``, with a
    scripted ``mlx_audio.audio_io.read`` (miniaudio is not in this image): (a) a signal longer than the window (2.3 windows of 0.1 s -> three chunks, the last
    one zero-padded), (b) a signal shorter than the window (one chunk, ``padding`` True); also what ``get_delay`` / ``get_output_length`` return on this
    architecture and a ``DACFile`` save / load round trip."""
    from mlx_audio_amd.codec.models.descript import make_dac_encoder_weights, make_dac_weights

    rd = _load_dac_modules()
    rb = sys.modules["mlx_audio.codec.models.descript.base"]
    c = DAC_ENC_TINY
    w = make_dac_weights(c["decoder_dim"], c["decoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_size"], c["codebook_dim"], seed=seed_w)
    w.update(make_dac_encoder_weights(c["encoder_dim"], c["encoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_dim"], seed=seed_w))
    model = rd.DAC(**c)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    model.eval()
    g = np.random.default_rng(seed_audio)
    sr = c["sample_rate"]
    out = dict(seed_w=seed_w, config=json.dumps(c), delay=np.int32(model.delay), out_len_1000=np.int32(model.get_output_length(1000)))
    fake = types.ModuleType("mlx_audio.audio_io")
    saved = sys.modules.get("mlx_audio.audio_io")
    try:
        for tag, n, win in (("long", int(0.23 * sr), 0.1), ("short", int(0.07 * sr), 0.1)):
            t = np.arange(n) / sr
            sig = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.05 * g.standard_normal(n)).astype(np.float32)
            fake.read = lambda path, _s=sig: (_s, sr)
            sys.modules["mlx_audio.audio_io"] = fake
            f = model.compress("scripted.wav", win_duration=win)
            rec = model.decompress(f)
            out.update({f"{tag}_signal": sig, f"{tag}_codes": np.asarray(f.codes).astype(np.int32), f"{tag}_chunk_length": np.int32(f.chunk_length),
                        f"{tag}_original_length": np.float64(f.original_length), f"{tag}_input_db": np.float32(np.asarray(f.input_db)), f"{tag}_padding": np.bool_(f.padding),
                        f"{tag}_channels": np.int32(f.channels), f"{tag}_recons": _np(rec), f"{tag}_win": np.float32(win)})
            if tag == "long":
                f2 = model.compress("scripted.wav", win_duration=win, n_quantizers=2)
                out["long_codes_nq2"] = np.asarray(f2.codes).astype(np.int32)
    finally:
        if saved is not None:
            sys.modules["mlx_audio.audio_io"] = saved
        else:
            sys.modules.pop("mlx_audio.audio_io", None)
    out["padding_after"] = np.bool_(model.padding)
    return out


def run_snac(seed_w, seed_codes, n_frames, attn_window_size=None):
    """The reference's ``SNAC.quantizer.from_codes`` + ``SNAC.decoder`` (codec/models/snac/{snac,layers,vq}.py), depthwise convs, no attention, with the
    NoiseBlock's gaussian draws logged."""
    from mlx_audio_amd.codec.models.snac import make_snac_weights

    _codec_pkgs()
    base = "mlx_audio.codec.models.snac"
    _pkg(base, f"{REF}/codec/models/snac")
    _load(f"{base}.attention", f"{REF}/codec/models/snac/attention.py")
    _load(f"{base}.layers", f"{REF}/codec/models/snac/layers.py")
    _load(f"{base}.vq", f"{REF}/codec/models/snac/vq.py")
    rsn = _load(f"{base}.snac", f"{REF}/codec/models/snac/snac.py")
    cfg = dict(sampling_rate=24000, encoder_dim=8, encoder_rates=[2, 4, 8], decoder_dim=128 if attn_window_size else 64, decoder_rates=[8, 4, 2],
               attn_window_size=attn_window_size, codebook_size=64, codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True)
    latent = cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
    w = make_snac_weights(latent, cfg["decoder_dim"], cfg["decoder_rates"], cfg["vq_strides"], cfg["codebook_size"], cfg["codebook_dim"], True, True, seed=seed_w,
                          attn=attn_window_size is not None)
    model = rsn.SNAC(**cfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    dec_missing = [m for m in missing if not m.startswith("encoder.") and ".in_proj." not in m and ".rel_pos.inv_freq" not in m]  # inv_freq: a constant the module builds itself
    assert not unexpected and not dec_missing and not mism, (dec_missing[:8], unexpected[:8], mism[:4])
    model.eval()
    g = np.random.default_rng(seed_codes)
    codes = [g.integers(0, cfg["codebook_size"], size=(2, n_frames // s_)).astype(np.int32) for s_ in cfg["vq_strides"]]
    mx.random.seed(seed_codes)
    z = model.quantizer.from_codes([mx.array(c) for c in codes])
    audio = np.asarray(model.decoder(z.moveaxis(1, 2)))  # SNAC.decode (snac.py:101-104)
    noises = [v for kind, shp, v in mx.random.log if kind == "normal"]
    out = dict(seed_w=seed_w, z=np.asarray(z).astype(np.float32), audio=audio.astype(np.float32), n_noise=len(noises))
    out.update({f"codes{i}": c for i, c in enumerate(codes)})
    out.update({f"noise{i}": n.astype(np.float32) for i, n in enumerate(noises)})
    return out, cfg


SNAC_ENC_TINY = dict(sampling_rate=24000, encoder_dim=8, encoder_rates=[2, 3, 8], decoder_dim=64, decoder_rates=[8, 3, 2], attn_window_size=None, codebook_size=64,
                     codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True)


def run_snac_encode(seed_w, seed_audio, n_samples, depthwise=True):
    """The reference's ``SNAC.encode`` and ``SNAC.__call__`` (snac.py:88-102: preprocess -> Encoder -> ResidualVectorQuantize.__call__, vq.py:10-113) on a
    seeded checkpoint: even and odd strides, multi-scale levels (strides 4 / 2 / 1), a length that needs the right padding; depthwise or dense k7 convs."""
    from mlx_audio_amd.codec.models.snac import make_snac_encoder_weights, make_snac_weights

    _codec_pkgs()
    base = "mlx_audio.codec.models.snac"
    _pkg(base, f"{REF}/codec/models/snac")
    _load(f"{base}.attention", f"{REF}/codec/models/snac/attention.py")
    _load(f"{base}.layers", f"{REF}/codec/models/snac/layers.py")
    _load(f"{base}.vq", f"{REF}/codec/models/snac/vq.py")
    rsn = _load(f"{base}.snac", f"{REF}/codec/models/snac/snac.py")
    cfg = dict(SNAC_ENC_TINY, depthwise=depthwise)
    latent = cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
    w = make_snac_weights(latent, cfg["decoder_dim"], cfg["decoder_rates"], cfg["vq_strides"], cfg["codebook_size"], cfg["codebook_dim"], True, depthwise, seed=seed_w)
    w.update(make_snac_encoder_weights(cfg["encoder_dim"], cfg["encoder_rates"], latent, cfg["vq_strides"], cfg["codebook_dim"], depthwise, seed=seed_w))
    model = rsn.SNAC(**cfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    missing = [m for m in missing if ".rel_pos.inv_freq" not in m]
    assert not unexpected and not missing and not mism, (missing[:8], unexpected[:8], mism[:4])
    model.eval()
    g = np.random.default_rng(seed_audio)
    t = np.arange(n_samples) / cfg["sampling_rate"]
    audio = np.stack([0.5 * np.sin(2 * np.pi * (200 + 110 * b) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t)) + 0.15 * g.standard_normal(n_samples) for b in range(2)])
    audio = audio[:, None, :].astype(np.float32)
    padded = model.preprocess(mx.array(audio))
    z = model.encoder(padded.moveaxis(1, 2))
    z_q, codes = model.quantizer(z)
    codes2 = model.encode(mx.array(audio))
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(codes, codes2))
    # SNAC.__call__ (snac.py:88-94) hands the quantizer's [B, D, T] output to the channels-last decoder WITHOUT the moveaxis that decode() applies
    # (snac.py:106): it raises unless T == D.  Recorded, not pinned.
    try:
        model(mx.array(audio))
        call_error = ""
    except Exception as e:  # noqa: BLE001
        call_error = f"{type(e).__name__}: {str(e)[:160]}"
    out = dict(seed_w=seed_w, config=json.dumps(cfg), audio=audio, padded_len=np.int32(padded.shape[-1]), z=_np(z), z_q=_np(z_q), call_error=call_error)
    out.update({f"codes{i}": np.asarray(c).astype(np.int32) for i, c in enumerate(codes)})
    return out


def run_snac_local_mha_probe():
    """The reference's SNAC with ``attn_window_size`` set (the 32 / 44 kHz models) CANNOT run its decoder: ``LocalMHA.__call__`` (attention.py:19-23) is a
    line-by-line transcription of the PyTorch module for [B, C, T] data (``B, C, T = x.shape``, ``x.moveaxis(1, 2)``) but the MLX decoder hands it
    channels-last [B, T, C] data, so its LayerNorm([C]) meets a last axis of length T.  Recorded here as evidence (the error the reference raises),
    because this package's LocalMHA implements what the module MEANS and therefore has no reference output to be pinned to."""
    try:
        run_snac(seed_w=6, seed_codes=2, n_frames=16, attn_window_size=4)
        out = dict(raised=False)
    except Exception as e:  # noqa: BLE001
        out = dict(raised=True, error_type=type(e).__name__, message=str(e)[:200], where="mlx_audio/codec/models/snac/attention.py:22 (self.norm(x.moveaxis(1, 2)))")
    with open(os.path.join(HERE, "ref_snac_local_mha_probe.json"), "w") as f:
        json.dump(out, f)
    return out


def run_vocos(seed_w, seed_audio):
    """The reference's ``Vocos`` mel model: ``MelSpectrogramFeatures`` -> ``VocosBackbone`` (ConvNeXt blocks) -> ``ISTFTHead`` (codec/models/vocos/{vocos,mel}.py)."""
    from mlx_audio_amd.codec.models.vocos import make_vocos_weights

    _codec_pkgs()
    base = "mlx_audio.codec.models.vocos"
    _pkg(base, f"{REF}/codec/models/vocos")
    enc = types.ModuleType("mlx_audio.codec.models.encodec")   # EnCodec features are another model (LSTM): not built, not on this path
    enc.Encodec = type("Encodec", (), {})
    sys.modules["mlx_audio.codec.models.encodec"] = enc
    u = sys.modules["mlx_audio.utils"]
    dsp = sys.modules["mlx_audio.dsp"]
    u.hanning, u.istft, u.mel_filters, u.stft = dsp.hanning, dsp.istft, dsp.mel_filters, dsp.stft
    _load(f"{base}.mel", f"{REF}/codec/models/vocos/mel.py")
    rv = _load(f"{base}.vocos", f"{REF}/codec/models/vocos/vocos.py")
    cfg = {"feature_extractor": {"class_path": "vocos.feature_extractors.MelSpectrogramFeatures",
                                 "init_args": {"sample_rate": 24000, "n_fft": 1024, "hop_length": 256, "n_mels": 100}},
           "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 64, "intermediate_dim": 128, "num_layers": 2}},
           "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 64, "n_fft": 1024, "hop_length": 256}}}
    w = make_vocos_weights(cfg, seed=seed_w)
    model = rv.Vocos.from_hparams(cfg)
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected and not mism, (missing[:8], unexpected[:8], mism[:4])
    model.eval()
    audio = np.random.default_rng(seed_audio).standard_normal(12_000).astype(np.float32)
    feats = np.asarray(model.feature_extractor(mx.array(audio)))
    out = np.asarray(model(mx.array(audio)))
    return dict(seed_w=seed_w, seed_audio=seed_audio, features=feats.astype(np.float32), audio=out.astype(np.float32)), cfg


def import_qwen3_model():
    """The reference's ``tts/models/qwen3_tts/qwen3_tts.py`` (``Model``) under its real name."""
    name = "mlx_audio.tts.models.qwen3_tts.qwen3_tts"
    if name in sys.modules:
        return sys.modules[name]
    import_lm_and_mimi()
    _load("mlx_audio.lm.sample_utils", f"{REF}/lm/sample_utils.py")
    _load("mlx_audio.tts.continuous", f"{REF}/tts/continuous.py")
    if "mlx_audio.tts.models.qwen3_tts.config" not in sys.modules:
        _pkg("mlx_audio.tts.models.qwen3_tts", f"{REF}/tts/models/qwen3_tts")
        _load("mlx_audio.tts.models.qwen3_tts.config", f"{REF}/tts/models/qwen3_tts/config.py")
    for m in ("talker", "speech_tokenizer", "speaker_encoder"):
        if f"mlx_audio.tts.models.qwen3_tts.{m}" not in sys.modules:
            _load(f"mlx_audio.tts.models.qwen3_tts.{m}", f"{REF}/tts/models/qwen3_tts/{m}.py")
    u = sys.modules["mlx_audio.utils"]
    u.load_audio = lambda *a, **k: None
    return _load(name, f"{REF}/tts/models/qwen3_tts/qwen3_tts.py")


def run_qwen3_inputs(seed):
    """The reference's prompt assembly of Qwen3-TTS -- ``Model._prepare_generation_inputs`` (qwen3_tts.py:326-484: chat template slices, tts pad / bos /
    eos embeddings, the codec prefix with / without a language id, dialect override, speaker embedding, instruct prefix, first text token, trailing text)
    and ``_prepare_batch_inputs`` (:486-604: left-padded prefill, tts_pad-padded trailing text, mask and padding metadata) -- with the talker's embedding
    tables replaced by seeded lookup tables and the tokenizer by ``pt_layouts.QwenCharTokenizer``; cases in ``pt_layouts.QWEN3_INPUT_CASES``."""
    import pt_layouts as PT

    q = import_qwen3_model()
    g = np.random.default_rng(seed)
    H = 8
    text_table = g.standard_normal((PT.QWEN3_TEXT_VOCAB, H)).astype(np.float32)
    codec_table = g.standard_normal((PT.QWEN3_CODEC_VOCAB, H)).astype(np.float32)

    class Talker:
        def get_text_embeddings(self):
            return lambda ids: mx.array(text_table)[ids]

        def text_projection(self, x):
            return x

        def get_input_embeddings(self):
            return lambda ids: mx.array(codec_table)[ids]

    class Host:
        _prepare_generation_inputs = q.Model._prepare_generation_inputs
        _prepare_batch_inputs = q.Model._prepare_batch_inputs
        tokenizer = PT.QwenCharTokenizer()
        config = PT.qwen3_input_config()
        talker = Talker()
        speaker_encoder = None

    host = Host()
    out = dict(seed=seed, text_table=text_table, codec_table=codec_table)
    for i, c in enumerate(PT.QWEN3_INPUT_CASES):
        e, tr, pad = host._prepare_generation_inputs(c["text"], language=c["language"], speaker=c["speaker"], instruct=c["instruct"])
        out[f"embeds{i}"], out[f"trailing{i}"], out[f"pad{i}"] = (np.asarray(v).astype(np.float32) for v in (e, tr, pad))
    b = PT.QWEN3_BATCH_CASE
    bi = host._prepare_batch_inputs(b["texts"], language=b["language"], speakers=b["speakers"], instructs=b["instructs"], return_metadata=True)
    out.update(batch_embeds=np.asarray(bi.input_embeds).astype(np.float32), batch_trailing=np.asarray(bi.trailing_text_hidden).astype(np.float32),
               batch_pad=np.asarray(bi.tts_pad_embed).astype(np.float32), batch_mask=np.asarray(bi.attention_mask).astype(np.float32),
               left_padding=np.array(bi.left_padding), prefill_lens=np.array(bi.prefill_lens), trailing_lens=np.array(bi.trailing_lens))
    return out


def run_qwen3_speaker_encoder(seed_w, seed_mel, frames):
    """The reference's ``Qwen3TTSSpeakerEncoder`` (speaker_encoder.py:232-313: TDNN -> three SE-Res2Net blocks -> MFA -> attentive statistics pooling ->
    fc) on a seeded tiny checkpoint (loaded in the module's own layout) and a seeded mel batch; also the reference's ``sanitize`` of the same
    parameters in their PyTorch form at widths where its shape heuristic is unambiguous (``ref_sanitize``-style summary)."""
    import pt_layouts as PT
    import torch

    from mlx_audio_amd.tts.models.qwen3_tts import speaker_encoder as SE

    import_qwen3_model()
    rs = sys.modules["mlx_audio.tts.models.qwen3_tts.speaker_encoder"]
    rc = sys.modules["mlx_audio.tts.models.qwen3_tts.config"]
    c = SE.tiny_speaker_encoder_config()
    w = SE.make_speaker_encoder_weights(c, seed=seed_w)
    from dataclasses import asdict

    model = rs.Qwen3TTSSpeakerEncoder(rc.Qwen3TTSSpeakerEncoderConfig(**asdict(c)))
    model.load_weights([(k, v.numpy()) for k, v in w.items()])
    missing, unexpected, mism = model._load_report
    assert not missing and not unexpected and not mism, (missing[:8], unexpected[:8], mism[:4])
    mels = SE.make_mels(2, frames, c.mel_dim, seed=seed_mel).numpy()
    emb = np.asarray(model(mx.array(mels))).astype(np.float32)
    # sanitize: widths > 64 so that (out, in, 1) kernels are recognised as PyTorch-form (qwen3_tts.py:123-157)
    c2 = SE.Qwen3TTSSpeakerEncoderConfig(mel_dim=80, enc_dim=72, enc_channels=[96, 96, 96, 192], enc_kernel_sizes=[5, 3, 3, 1], enc_dilations=[1, 2, 3, 1],
                                         enc_attention_channels=80, enc_res2net_scale=4, enc_se_channels=72)
    w2 = SE.make_speaker_encoder_weights(c2, seed=seed_w + 1)
    ck = {"speaker_encoder." + k: (v.permute(0, 2, 1).contiguous() if v.dim() == 3 else v) for k, v in w2.items()}
    ck["talker.model.norm.weight"] = torch.ones(4)   # dropped: not a speaker-encoder key
    san = rs.Qwen3TTSSpeakerEncoder.sanitize({k: mx.array(v.numpy()) for k, v in ck.items()})
    import json

    return dict(seed_w=seed_w, seed_mel=seed_mel, frames=frames, embedding=emb,
                sanitize=json.dumps(PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})))


def run_qwen3_icl(seed):
    """The reference's voice-cloning host logic of Qwen3-TTS on scripted parts (``pt_layouts``: character tokenizer, seeded embedding tables, stand-ins for
    the speech tokenizer's encode / decode and for the x-vector): ``_prepare_icl_generation_inputs`` (qwen3_tts.py:606-803), the in-context branch of
    ``_prepare_batch_inputs`` (:509-527), ``_prepare_generation_inputs`` with a clip but no transcript (:383-384), ``_decode_icl_generated_codes``
    (:1085-1112), ``_normalize_shared_batch_refs`` (:1582-1649) and ``supports_tts_batch`` with references (:233-244)."""
    import json

    import pt_layouts as PT

    q = import_qwen3_model()
    g = np.random.default_rng(seed)
    H = 8
    text_table = g.standard_normal((PT.QWEN3_TEXT_VOCAB, H)).astype(np.float32)
    codec_table = g.standard_normal((PT.QWEN3_CODEC_VOCAB, H)).astype(np.float32)
    cp_tables = g.standard_normal((PT.QWEN3_ICL_GROUPS - 1, PT.QWEN3_ICL_CP_VOCAB, H)).astype(np.float32)

    class CodePredictor:
        codec_embedding = [(lambda ids, i=i: mx.array(cp_tables[i])[ids]) for i in range(PT.QWEN3_ICL_GROUPS - 1)]

    class Talker:
        code_predictor = CodePredictor()

        def get_text_embeddings(self):
            return lambda ids: mx.array(text_table)[ids]

        def text_projection(self, x):
            return x

        def get_input_embeddings(self):
            return lambda ids: mx.array(codec_table)[ids]

    class SpeechTokenizer:
        has_encoder = True

        def encode(self, audio):
            return mx.array(PT.qwen3_fake_codes(np.asarray(audio)))

        def decode(self, codes):
            a, n = PT.qwen3_fake_decode(np.asarray(codes))
            return mx.array(a), mx.array(n)

    class Host:
        _prepare_generation_inputs = q.Model._prepare_generation_inputs
        _prepare_icl_generation_inputs = q.Model._prepare_icl_generation_inputs
        _prepare_batch_inputs = q.Model._prepare_batch_inputs
        _decode_icl_generated_codes = q.Model._decode_icl_generated_codes
        _normalize_shared_batch_refs = q.Model._normalize_shared_batch_refs
        _same_shared_ref_value = staticmethod(q.Model._same_shared_ref_value)
        supports_tts_batch = q.Model.supports_tts_batch
        tokenizer = PT.QwenCharTokenizer()
        talker = Talker()
        sample_rate = 24000

        def __init__(self, xvec=True, kind="base", enc=True):
            self.config = PT.qwen3_icl_config(kind)
            self.speaker_encoder = object() if xvec else None
            self.speech_tokenizer = None if enc is None else SpeechTokenizer()
            if enc is False:
                self.speech_tokenizer.has_encoder = False
            self._icl_cache = {}

        def extract_speaker_embedding(self, audio, sr=24000):
            return mx.array(PT.qwen3_fake_xvector(np.asarray(audio), H))

    def clip(spec):
        n, s, lead = spec
        a = PT.qwen3_fake_clip(n, s)
        return mx.array(a.reshape((1,) * lead + (n,)))

    out = dict(seed=seed, text_table=text_table, codec_table=codec_table, cp_tables=cp_tables)
    hosts = {True: Host(True), False: Host(False)}
    for i, c in enumerate(PT.QWEN3_ICL_CASES):
        e, tr, pad, rc_ = hosts[c["xvec"]]._prepare_icl_generation_inputs(c["text"], ref_audio=clip(c["clip"]), ref_text=c["ref_text"], language=c["language"])
        out[f"icl_embeds{i}"], out[f"icl_trailing{i}"], out[f"icl_pad{i}"] = (np.asarray(v).astype(np.float32) for v in (e, tr, pad))
        out[f"icl_codes{i}"] = np.asarray(rc_).astype(np.int64)
    out["cache_entries"] = np.array([len(hosts[True]._icl_cache), len(hosts[False]._icl_cache)])
    b = PT.QWEN3_ICL_BATCH
    bi = Host(True)._prepare_batch_inputs(b["texts"], language=b["language"], ref_audio=clip(b["clip"]), ref_text=b["ref_text"], return_metadata=True)
    out.update(batch_embeds=np.asarray(bi.input_embeds).astype(np.float32), batch_trailing=np.asarray(bi.trailing_text_hidden).astype(np.float32),
               batch_mask=np.asarray(bi.attention_mask).astype(np.float32), left_padding=np.array(bi.left_padding), prefill_lens=np.array(bi.prefill_lens),
               trailing_lens=np.array(bi.trailing_lens), batch_ref_codes=np.asarray(bi.ref_codes).astype(np.int64))
    for i, c in enumerate(PT.QWEN3_XVEC_CASES):
        e, tr, pad = Host(c["xvec"])._prepare_generation_inputs(c["text"], language=c["language"], speaker=c["speaker"], ref_audio=clip(c["clip"]))
        out[f"xvec_embeds{i}"], out[f"xvec_trailing{i}"] = np.asarray(e).astype(np.float32), np.asarray(tr).astype(np.float32)
    h = Host(True)
    for i, (n_gen, n_ref) in enumerate(PT.QWEN3_ICL_DECODE_CASES):
        gen, ref = PT.qwen3_icl_decode_case(n_gen, n_ref)
        audio = h._decode_icl_generated_codes([mx.array(gen[j:j + 1]) for j in range(n_gen)], mx.array(ref))
        out[f"decoded{i}"] = np.asarray(audio).astype(np.float32)
    q.load_audio = lambda path, sample_rate=None: "loaded:" + str(path)
    tables = dict(shared=[PT.qwen3_shared_ref_outcome(h._normalize_shared_batch_refs, c) for c in PT.QWEN3_SHARED_REF_CASES],
                  supports=[bool(Host(True, c["kind"], c["enc"]).supports_tts_batch(**c["kw"])) for c in PT.QWEN3_SUPPORTS_BATCH_CASES])
    out["tables"] = json.dumps(tables)
    return out


def run_kokoro_pipeline():
    """The reference's ``KokoroPipeline`` (tts/models/kokoro/pipeline.py) with the G2P and the model replaced by stand-ins: ``__call__`` on English
    token streams (``en_tokenize`` / ``waterfall_last`` / ``tokens_to_ps`` / ``tokens_to_text``, ``join_timestamps`` on the stand-in's durations,
    :231-294, 360-399, 444-470) and on other languages (400-character text chunks, truncation at 510 phonemes, :473-528), and ``generate_from_tokens``
    on a token list; cases in ``pt_layouts.KOKORO_PIPELINE_CASES``."""
    import json

    import pt_layouts as PT

    _pkg("mlx_audio.tts.models.kokoro", f"{REF}/tts/models/kokoro")
    if "mlx_audio.tts.models.kokoro.voice" not in sys.modules:
        _load("mlx_audio.tts.models.kokoro.voice", f"{REF}/tts/models/kokoro/voice.py")
    hub = sys.modules.get("huggingface_hub")
    rp = _load("mlx_audio.tts.models.kokoro.pipeline", f"{REF}/tts/models/kokoro/pipeline.py")
    out = []
    for case in PT.KOKORO_PIPELINE_CASES:
        calls = []

        def model(ps, ref_s, speed, return_output=True, calls=calls):
            calls.append(dict(ps=ps, row=int(np.asarray(ref_s)[0]), speed=float(speed)))
            return types.SimpleNamespace(audio=mx.zeros((len(ps) * 10,)), pred_dur=mx.array(np.array(PT.kokoro_fake_durations(ps), dtype=np.int32)))

        pipe = rp.KokoroPipeline.__new__(rp.KokoroPipeline)
        pipe.lang_code, pipe.repo_id, pipe.model = case["lang"], "repo", model
        pipe.voices = {"v": mx.array(np.arange(512, dtype=np.float32)[:, None] * np.ones((1, 4), dtype=np.float32))}
        if "tokens" in case:
            toks = PT.kokoro_token_stream(*case["tokens"])
            pipe.g2p = lambda text, toks=toks: ("", toks)
            results = list(pipe("some text", voice="v", speed=1.25))
        else:
            pipe.g2p = PT.kokoro_spanish_g2p
            kw = {"split_pattern": case["split_pattern"]} if "split_pattern" in case else {}
            results = list(pipe(case["text"], voice="v", speed=0.9, **kw))
        rec = [dict(graphemes=r.graphemes, phonemes=r.phonemes, text_index=r.text_index,
                    tokens=None if r.tokens is None else [[t.text, t.phonemes, t.start_ts, t.end_ts] for t in r.tokens]) for r in results]
        entry = dict(name=case["name"], results=rec, calls=calls)
        if case["name"] == "en_mixed":   # the same stream through generate_from_tokens (fresh tokens: en_tokenize rewrites them in place)
            calls2 = []
            pipe.model = lambda ps, ref_s, speed, return_output=True: model(ps, ref_s, speed, calls=calls2)
            res2 = list(pipe.generate_from_tokens(PT.kokoro_token_stream(*case["tokens"]), voice="v", speed=1.0))
            entry["from_tokens"] = [dict(graphemes=r.graphemes, phonemes=r.phonemes, n_tokens=len(r.tokens), last_end=r.tokens[-1].end_ts) for r in res2]
            entry["from_tokens_calls"] = calls2
        out.append(entry)
    with open(os.path.join(HERE, "ref_kokoro_pipeline.json"), "w") as f:
        json.dump(out, f)
    return [(e["name"], [len(r["phonemes"]) for r in e["results"]]) for e in out]


def run_broker():
    """The reference's ``InferenceBroker`` (server_inference.py:129-358) driven by ``pt_layouts.broker_scenario``."""
    import json

    import pt_layouts as PT

    mod = _load("mlx_audio.server_inference", f"{REF}/server_inference.py")
    out = PT.broker_scenario(mod)
    with open(os.path.join(HERE, "ref_broker.json"), "w") as f:
        json.dump(out, f)
    return out["trace"]


def run_qwen3_session():
    """The reference's ``Qwen3TTSBatchSession`` (continuous_batching.py:37-360) over a SCRIPTED model (every request's EOS frame is fixed by
    ``pt_layouts.QWEN3_SESSION``; the talker only moves the reference's own KV caches along): which requests each ``step()`` advances and admits,
    and the events it returns, for arrivals / cancellations in the middle of other requests' utterances."""
    import json
    from types import SimpleNamespace

    import pt_layouts as PT

    if "mlx_audio" not in sys.modules:
        import_reference()
    import_lm_and_mimi()   # lm/models/cache.py: the session's KVCache / BatchKVCache
    if "mlx_audio.tts.continuous" not in sys.modules:
        _load("mlx_audio.tts.continuous", f"{REF}/tts/continuous.py")
    if "mlx_audio.tts.models.qwen3_tts" not in sys.modules:
        _pkg("mlx_audio.tts.models.qwen3_tts", f"{REF}/tts/models/qwen3_tts")
    mod = _load("mlx_audio.tts.models.qwen3_tts.continuous_batching", f"{REF}/tts/models/qwen3_tts/continuous_batching.py")
    cont = sys.modules["mlx_audio.tts.continuous"]
    cfg = PT.QWEN3_SESSION
    EOS, script = cfg["eos"], cfg["script"]

    def ids_of(x):  # channel 0 of the last position carries the sequence id
        return [int(round(float(v))) for v in np.asarray(x)[:, -1, 0]]

    def embed_for_tokens(code_tokens):
        tok = np.asarray(code_tokens[0]).reshape(-1)
        e = np.zeros((tok.shape[0], 1, 4), dtype=np.float32)
        e[:, 0, 0] = tok // 1000
        return mx.array(e)

    class Model:
        sample_rate = 24000
        speech_tokenizer = object()
        config = SimpleNamespace(talker_config=SimpleNamespace(codec_eos_token_id=EOS, num_hidden_layers=1, vocab_size=3000))

        def _suppress_codec_tokens(self, eos):
            return []

        def _prepare_batch_inputs(self, texts, language="auto", speakers=None, instructs=None, return_metadata=False):
            ids = [int(t) for t in texts]
            plens = [2 + (i % 3) for i in ids]
            L = max(plens)
            x = np.zeros((len(ids), L, 4), dtype=np.float32)
            mask = np.zeros((len(ids), L), dtype=np.float32)
            for b, (i, n) in enumerate(zip(ids, plens)):
                x[b, L - n:, 0] = i
                mask[b, L - n:] = 1
            tl = [1 + (i % 2) for i in ids]
            return SimpleNamespace(input_embeds=mx.array(x), attention_mask=mx.array(mask), left_padding=[L - n for n in plens],
                                   trailing_text_hidden=mx.zeros((len(ids), max(tl), 4)), tts_pad_embed=mx.zeros((1, 1, 4)), trailing_lens=tl)

        def talker(self, x, cache=None, attention_mask=None):
            B, L, _ = x.shape
            for c in cache:
                c.update_and_fetch(mx.zeros((B, 1, L, 2)), mx.zeros((B, 1, L, 2)))
            return x[:, -1:, :2], x[:, -1:, :]

        def _sample_token_batch(self, logits, temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.05, generated_tokens_per_seq=None,
                                suppress_tokens=None, min_p=0.0):
            out = []
            for b, i in enumerate(ids_of(logits)):
                f = len(generated_tokens_per_seq[b])
                out.append([EOS if f >= script[i] else 1000 * i + f])
            return mx.array(np.asarray(out, dtype=np.int32))

        def _predict_code_tokens(self, first_token, hidden, *, temperature, top_k, top_p, code_cache=None):
            return [first_token], first_token

        def _next_batch_input_embeds(self, trailing, pad, indices, code_tokens, *, pad_when_index_clamped=False):
            return embed_for_tokens(code_tokens)

        def _codec_embeds_for_tokens(self, code_tokens):
            return embed_for_tokens(code_tokens)

        def _decode_generated_codes(self, codes):
            return mx.zeros((10 * len(codes),))

    session = mod.Qwen3TTSBatchSession(Model(), cont.TTSBatchOptions(temperature=0.0, max_tokens=cfg["max_tokens"], max_batch_size=cfg["max_batch"]))
    trace = []
    adv, adm = session._advance_active, session._take_pending_batch

    def advance():
        trace.append(["advance", [st.sequence_id for st in session._active]])
        return adv()

    def take():
        batch = adm()
        if batch:
            trace.append(["admit", [it.sequence_id for it in batch]])
        return batch

    session._advance_active, session._take_pending_batch = advance, take
    rows = PT.qwen3_session_drive(session, lambda i: cont.TTSBatchItem(sequence_id=i, text=str(i)))
    out = dict(rows=rows, trace=trace)
    with open(os.path.join(HERE, "ref_qwen3_session.json"), "w") as f:
        json.dump(out, f)
    return dict(steps=len(rows), calls=len(trace))


ENCODEC_TINY = dict(audio_channels=1, num_filters=8, kernel_size=7, num_residual_layers=1, dilation_growth_rate=2, codebook_size=64, codebook_dim=32,
                    hidden_size=32, num_lstm_layers=2, residual_kernel_size=3, use_causal_conv=True, normalize=False, pad_mode="reflect",
                    norm_type="weight_norm", last_kernel_size=7, trim_right_ratio=1.0, compress=2, upsampling_ratios=[4, 2, 2],
                    target_bandwidths=[15.0, 60.0], sampling_rate=24000)


def run_encodec(seed_w, seed_codes, n_frames):
    """The reference's ``Encodec.decode`` (codec/models/encodec/encodec.py:740-777 -> _decode_frame -> RVQ decode -> EncodecDecoder with its LSTM) on a
    seeded checkpoint of a small config (causal, reflect padding, three upsampling stages, two LSTM layers), batch 1 (see mlx_shim._metal_kernel)."""
    if "mlx_audio" not in sys.modules:
        import_reference()
    if "mlx_audio.codec" not in sys.modules:
        _pkg("mlx_audio.codec", f"{REF}/codec")
        _pkg("mlx_audio.codec.models", f"{REF}/codec/models")
    _pkg("mlx_audio.codec.models.encodec", f"{REF}/codec/models/encodec")
    E = _load("mlx_audio.codec.models.encodec.encodec", f"{REF}/codec/models/encodec/encodec.py")
    sys.path.insert(0, ROOT)
    from mlx_audio_amd.codec.models.encodec.encodec import make_encodec_weights

    cfg = E.EncodecConfig(**ENCODEC_TINY)
    model = E.Encodec(cfg)
    w = make_encodec_weights(ENCODEC_TINY, seed=seed_w)
    missing = model.load_weights([(k, mx.array(v.numpy())) for k, v in w.items() if k.startswith(("decoder.", "quantizer."))], strict=False)
    nq = model.quantizer.num_quantizers
    g = np.random.default_rng(seed_codes)
    codes = g.integers(0, ENCODEC_TINY["codebook_size"], size=(1, 1, nq, n_frames)).astype(np.int32)   # [B, chunks = 1, nq, T]
    audio = model.decode(mx.array(codes), [None])
    emb = model.quantizer.decode(mx.array(codes[:, 0]))
    np.savez_compressed(os.path.join(HERE, "ref_encodec_tiny.npz"), seed_w=seed_w, seed_codes=seed_codes, n_frames=n_frames, codes=codes,
                        config=json.dumps(ENCODEC_TINY), audio=_np(audio), embeddings=_np(emb))
    return dict(audio=list(np.asarray(audio).shape), nq=int(nq), peak=float(np.abs(np.asarray(audio)).max()), missing=str(missing)[:80])


ENCODEC_ENC_STEREO = dict(audio_channels=2, num_filters=8, kernel_size=7, num_residual_layers=1, dilation_growth_rate=2, codebook_size=64, codebook_dim=32,
                          hidden_size=32, num_lstm_layers=1, residual_kernel_size=3, use_causal_conv=False, normalize=True, pad_mode="reflect",
                          norm_type="weight_norm", last_kernel_size=7, trim_right_ratio=1.0, compress=2, upsampling_ratios=[5, 2, 2],
                          target_bandwidths=[18.0, 60.0], sampling_rate=24000, chunk_length_s=0.05, overlap=0.2)


def run_encodec_encode(seed_w, seed_audio, cfg_dict, n_samples, tag):
    """The reference's ``Encodec.encode`` (encodec.py:585-650 -> _encode_frame -> EncodecEncoder -> quantizer.encode) on a seeded checkpoint, batch 1 (the
    Metal LSTM kernel's index arithmetic, see mlx_shim._metal_kernel): (a) mono, causal, one chunk, both bandwidths; (b) stereo, non-causal, an odd stride, loudness normalisation, overlapping chunks -- input made by the reference's own ``preprocess_audio``."""
    if "mlx_audio" not in sys.modules:
        import_reference()
    if "mlx_audio.codec" not in sys.modules:
        _pkg("mlx_audio.codec", f"{REF}/codec")
        _pkg("mlx_audio.codec.models", f"{REF}/codec/models")
    _pkg("mlx_audio.codec.models.encodec", f"{REF}/codec/models/encodec")
    E = _load("mlx_audio.codec.models.encodec.encodec", f"{REF}/codec/models/encodec/encodec.py")
    sys.path.insert(0, ROOT)
    from mlx_audio_amd.codec.models.encodec.encodec import make_encodec_encoder_weights, make_encodec_weights

    cfg = E.EncodecConfig(**cfg_dict)
    model = E.Encodec(cfg)
    w = make_encodec_weights(cfg_dict, seed=seed_w)
    w.update(make_encodec_encoder_weights(cfg_dict, seed=seed_w))
    model.load_weights([(k, mx.array(v.numpy())) for k, v in w.items()], strict=True)
    g = np.random.default_rng(seed_audio)
    t = np.arange(n_samples) / cfg_dict["sampling_rate"]
    ch = cfg_dict["audio_channels"]
    raw = np.stack([0.5 * np.sin(2 * np.pi * (210 + 130 * c) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 9 * t)) + 0.15 * g.standard_normal(n_samples) for c in range(ch)], axis=1)
    raw = raw.astype(np.float32)
    inputs, masks = E.preprocess_audio(mx.array(raw if ch > 1 else raw[:, 0]), cfg_dict["sampling_rate"], model.chunk_length, model.chunk_stride)
    out = dict(seed_w=seed_w, config=json.dumps(cfg_dict), raw=raw, inputs=_np(inputs), masks=np.asarray(masks).astype(np.bool_))
    emb = model.encoder(inputs[:, :model.chunk_length] if model.chunk_length else inputs)
    out["embeddings_chunk0_unnormalised"] = _np(emb)
    for bw in cfg_dict["target_bandwidths"]:
        codes, scales = model.encode(inputs, masks, bandwidth=bw)
        out[f"codes_bw{bw}"] = np.asarray(codes).astype(np.int32)
        out[f"scales_bw{bw}"] = np.stack([_np(sc) for sc in scales]) if scales[0] is not None else np.zeros(0, np.float32)
    audio = model.decode(codes, scales, masks)
    out["decoded"] = _np(audio)
    np.savez_compressed(os.path.join(HERE, f"ref_encodec_encode_{tag}.npz"), **out)
    return {a: (v.shape if hasattr(v, "shape") else v) for a, v in out.items() if a != "config"}


def run_dataclasses(R):
    """Field names and defaults of the record types that cross the boundary: ``GenerationResult`` / ``BatchGenerationResult`` (tts/models/base.py),
    ``TTSBatchOptions / Item / Event`` (tts/continuous.py), the broker's request / context / chunk records (server_inference.py), Whisper's
    ``DecodingOptions`` / ``DecodingResult`` (decoding.py), ``STTOutput`` (stt/models/base.py)."""
    import dataclasses
    import json

    def fields(cls):
        out = []
        for f in dataclasses.fields(cls):
            d = f.default if f.default is not dataclasses.MISSING else ("<factory>" if f.default_factory is not dataclasses.MISSING else "<required>")
            out.append([f.name, d if isinstance(d, (int, float, str, bool, type(None))) else repr(d)])
        return out

    import_qwen3_model()
    _, dec = import_whisper()
    base = sys.modules["mlx_audio.tts.models.base"]
    cont = sys.modules["mlx_audio.tts.continuous"]
    srv = sys.modules.get("mlx_audio.server_inference") or _load("mlx_audio.server_inference", f"{REF}/server_inference.py")
    stt = sys.modules.get("mlx_audio.stt.models.base") or _load("mlx_audio.stt.models.base", f"{REF}/stt/models/base.py")
    out = {"GenerationResult": fields(base.GenerationResult), "BatchGenerationResult": fields(base.BatchGenerationResult),
           "TTSBatchOptions": fields(cont.TTSBatchOptions), "TTSBatchItem": fields(cont.TTSBatchItem), "TTSBatchEvent": fields(cont.TTSBatchEvent),
           "InferenceResultChunk": fields(srv.InferenceResultChunk), "InferenceContext": fields(srv.InferenceContext),
           "InferenceRequest": [f for f in fields(srv.InferenceRequest)], "DecodingOptions": fields(dec.DecodingOptions),
           "DecodingResult": fields(dec.DecodingResult), "STTOutput": fields(stt.STTOutput)}
    with open(os.path.join(HERE, "ref_dataclasses.json"), "w") as f:
        json.dump(out, f, indent=0)
    return {k: len(v) for k, v in out.items()}


def run_sampler(seed):
    """The reference's sampling chain: ``Model._sample_token_batch`` of Qwen3-TTS (suppress ids, per-sequence repetition penalty, temperature, top-k,
    top-p / min-p through ``lm/sample_utils.py``; qwen3_tts.py:862-925) with the final ``categorical_sampling`` replaced by a probe that records the
    filtered logits it was handed -- i.e. everything up to the random draw, for five parameter sets."""
    q = import_qwen3_model()
    g = np.random.default_rng(seed)
    B, V = 3, 257
    logits = (g.standard_normal((B, 1, V)) * 3).astype(np.float32)
    hist = [list(map(int, g.integers(0, V, size=n))) for n in (0, 5, 40)]
    sup = [V - 7, V - 3, 11]
    cases = [dict(temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.05), dict(temperature=0.7, top_k=0, top_p=0.8, repetition_penalty=1.0),
             dict(temperature=1.0, top_k=20, top_p=0.6, repetition_penalty=1.3), dict(temperature=1.3, top_k=0, top_p=1.0, repetition_penalty=1.1, min_p=0.05),
             dict(temperature=0.0, top_k=50, top_p=1.0, repetition_penalty=1.5)]
    out = {}
    seen = []
    orig = q.categorical_sampling
    q.categorical_sampling = lambda lg, temp: (seen.append(np.asarray(lg)), mx.argmax(lg, axis=-1))[1]
    try:
        for i, kw in enumerate(cases):
            seen.clear()
            tok = q.Model._sample_token_batch(None, mx.array(logits), generated_tokens_per_seq=hist, suppress_tokens=sup, **kw)
            out[f"tok{i}"] = np.asarray(tok).astype(np.int32)
            if seen:
                out[f"filtered{i}"] = seen[0].astype(np.float32)
    finally:
        q.categorical_sampling = orig
    import json

    return dict(seed=seed, logits=logits, hist=json.dumps(hist), suppress=np.array(sup, dtype=np.int32), cases=json.dumps(cases), **out)


def run_dsp(R, seed):
    """The reference's ``dsp.py`` entry points of the path (stft / istft with both normalisations / mel_filters / ISTFTCache.istft / compute_fbank_kaldi with
    dither 0) and Qwen3-TTS's ``mel_spectrogram`` (qwen3_tts.py:64-120), on seeded noise + tones."""
    dsp = R["dsp"]
    g = np.random.default_rng(seed)
    t = np.arange(12000) / 24000.0
    x = (0.2 * g.standard_normal(12000) + 0.4 * np.sin(2 * np.pi * 330 * t) + 0.1 * np.sin(2 * np.pi * 5000 * t)).astype(np.float32)
    out = dict(seed=seed)
    s1 = dsp.stft(mx.array(x), n_fft=400, hop_length=160, window=dsp.hanning(400))
    out["stft_400_160"] = np.asarray(s1).astype(np.complex64)
    s2 = dsp.stft(mx.array(x), n_fft=1024, hop_length=256, win_length=1024, window="hann", center=True, pad_mode="constant")
    out["stft_1024_256_constant"] = np.asarray(s2).astype(np.complex64)
    for norm in (False, True):
        y = dsp.istft(s2.T if hasattr(s2, "T") else s2.transpose(1, 0), hop_length=256, win_length=1024, window="hann", center=True, length=12000, normalized=norm)
        out[f"istft_norm{int(norm)}"] = np.asarray(y).astype(np.float32)
    out["mel_slaney"] = np.asarray(dsp.mel_filters(16000, 400, 80, norm="slaney", mel_scale=None)).astype(np.float32)
    out["mel_htk"] = np.asarray(dsp.mel_filters(24000, 1024, 128, f_min=0, f_max=12000, norm=None, mel_scale="htk")).astype(np.float32)
    cache = dsp.ISTFTCache()
    spec = np.asarray(s2).T[None]  # [1, bins, frames]
    w = dsp.hanning(1024 + 1)[:-1]
    yc = cache.istft(mx.array(spec.real.astype(np.float32)), mx.array(spec.imag.astype(np.float32)), 1024, 256, 1024, w, center=True, audio_length=12000)
    out["istft_cache"] = np.asarray(yc).astype(np.float32)
    x48 = np.concatenate([x, x, x, x])[:40000]
    fb = dsp.compute_fbank_kaldi(mx.array(x48[None, :]), sample_rate=48000, win_len=1920, win_inc=384, num_mels=60, win_type="hamming", dither=0.0)
    out["fbank"] = np.asarray(fb).astype(np.float32)
    q = sys.modules["mlx_audio.tts.models.qwen3_tts.qwen3_tts"]
    out["qwen3_mel"] = np.asarray(q.mel_spectrogram(mx.array(x))).astype(np.float32)
    return out


def run_sanitize(R):
    """The reference's ``sanitize`` of Kokoro (kokoro.py:178-275 + istftnet.py:998-1011) and CSM (sesame.py:577-604) on checkpoints in their published
    on-disk form (tests/golden/pt_layouts.py): resulting key -> [shape, sum, sum of squares]."""
    import json

    import pt_layouts as PT
    import torch

    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from mlx_audio_amd.tts.models.sesame import engine as E

    out = {}
    cfg = S.tiny_config()
    K = R["kokoro"]
    model = K.Model(K.ModelConfig.from_dict(cfg), repo_id="none")
    ck = PT.kokoro_checkpoint(S.make_kokoro_weights(cfg, seed=1))
    san = model.sanitize({k: mx.array(v.numpy()) for k, v in ck.items()})
    out["kokoro"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    rs = import_sesame()
    ck = PT.csm_checkpoint(E.make_csm_weights(E.tiny_csm(), seed=5))
    san = rs.Model.sanitize(None, {k: mx.array(v.numpy()) for k, v in ck.items()})
    out["csm"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    from mlx_audio_amd.tts.models.qwen3_tts import synthetic as QS

    if "mlx_audio.tts.models.qwen3_tts.qwen3_tts" not in sys.modules:
        run_sampler(0)  # loads the qwen3_tts modules
    st = sys.modules["mlx_audio.tts.models.qwen3_tts.speech_tokenizer"]
    ck = PT.qwen3_codec_checkpoint(QS.make_codec_decoder_weights(QS.tiny_codec_config(), seed=4))
    san = st.Qwen3TTSSpeechTokenizer.sanitize({k: mx.array(v.numpy()) for k, v in ck.items()})
    out["qwen3_codec"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    # ... and the ENCODER half of the same checkpoint (speech_tokenizer.py:1229-1381, 1418-1441): SeanetEncoder layer indices, q | k | v -> in_proj, embed_sum codebooks
    from mlx_audio_amd.codec.models.mimi import mimi as MM

    mc = MM.tiny_mimi_config()
    mw = {**MM.make_mimi_decoder_weights(mc, seed=8), **MM.make_mimi_encoder_weights(mc, seed=8)}
    ck = PT.qwen3_tokenizer_encoder_checkpoint(mw, mc.num_layers, mc.quantizer_nq)
    san = st.Qwen3TTSSpeechTokenizer.sanitize({k: mx.array(v.numpy()) for k, v in ck.items()})
    out["qwen3_tokenizer_encoder"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    # Whisper from the HF hub (whisper.py:551-617), Qwen3-TTS Model.sanitize with its layout heuristic (qwen3_tts.py:123-157, 2914-2937), KittenTTS's
    # Snake parameter names (kitten_tts.py:394-404)
    from mlx_audio_amd.stt.models.whisper import synthetic as WS

    wh, _ = import_whisper()
    ck = PT.whisper_hf_checkpoint(WS.make_whisper_weights(WS.tiny_dims(), seed=1))
    san = wh.Model.sanitize(types.SimpleNamespace(dtype=mx.float32), {k: mx.array(v.numpy()) for k, v in ck.items()})
    out["whisper_hf"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    q = import_qwen3_model()
    san = q.Model.sanitize({k: mx.array(v.numpy()) for k, v in PT.qwen3_model_checkpoint(3).items()})
    out["qwen3_model"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    for i, ck in enumerate(PT.kitten_alpha_checkpoints()):
        san = R["kitten"].Model.sanitize(None, {k: mx.array(v.numpy()) for k, v in ck.items()})
        out[f"kitten_alpha{i}"] = PT.summary({k: torch.from_numpy(np.asarray(v)) for k, v in san.items()})
    with open(os.path.join(HERE, "ref_sanitize.json"), "w") as f:
        json.dump(out, f)
    return {k: len(v) for k, v in out.items()}


BIGVGAN_TINY = dict(num_mels=20, upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=64, resblock="1",
                    resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], activation="snakebeta", snake_logscale=True,
                    use_bias_at_final=True, use_tanh_at_final=True)


def run_bigvgan(seed_w, seed_mel, n_frames):
    """The reference's ``BigVGAN`` (codec/models/bigvgan/{bigvgan,amp,resample,activation,conv}.py): mel -> waveform, AMPBlock1 and AMPBlock2 variants."""
    from mlx_audio_amd.codec.models.bigvgan import BigVGANConfig, make_bigvgan_weights

    _codec_pkgs()
    base = "mlx_audio.codec.models.bigvgan"
    if base not in sys.modules:
        _pkg(base, f"{REF}/codec/models/bigvgan")
        for m in ("activation", "conv", "resample", "amp", "bigvgan"):
            _load(f"{base}.{m}", f"{REF}/codec/models/bigvgan/{m}.py")
    rb = sys.modules[f"{base}.bigvgan"]
    out = dict(seed_w=seed_w, seed_mel=seed_mel, n_frames=n_frames)
    mel = (np.random.default_rng(seed_mel).standard_normal((2, BIGVGAN_TINY["num_mels"], n_frames)) * 0.8).astype(np.float32)
    for kind in ("1", "2"):
        cfgd = dict(BIGVGAN_TINY, resblock=kind)
        w = make_bigvgan_weights(BigVGANConfig(**cfgd), seed=seed_w)
        model = rb.BigVGAN(rb.BigVGANConfig(**cfgd))
        model.load_weights([(k, v.numpy()) for k, v in w.items()])
        missing, unexpected, mism = model._load_report
        assert not missing and not unexpected and not mism, (missing[:8], unexpected[:8], mism[:4])
        model.eval()
        out[f"audio{kind}"] = np.asarray(model(mx.array(mel))).astype(np.float32)
    return out


def run_cache():
    """The reference's own ``KVCache`` / ``BatchKVCache`` (lm/models/cache.py:104-176, 502-717) through the operation sequences tests/test_cache_cpu.py
    runs on this package's device-layout mirrors: offsets, capacities, left paddings and the fetched contents (as checksums)."""
    import json

    import_lm_and_mimi()
    C = sys.modules["mlx_audio.lm.models.cache"]
    G, DH = 2, 4
    W = G * DH

    def kv(B, n, seed):  # the same draws as tests/test_cache_cpu.py::_kv, in the reference's [B, heads, n, dh] layout
        import torch

        g = torch.Generator().manual_seed(seed)
        k, v = torch.randn(B, n, W, generator=g), torch.randn(B, n, W, generator=g)
        to = lambda t: mx.array(t.reshape(B, n, G, DH).transpose(1, 2).contiguous().numpy())  # noqa: E731
        return to(k), to(v)

    def chk(a):
        a = np.asarray(a, dtype=np.float64)
        return [list(a.shape), float(a.sum()), float((a ** 2).sum())]

    out = {}
    c = C.KVCache()
    rows = []
    for i, n in enumerate((3, 1, 1, 260, 1, 300)):
        k, v = kv(2, n, i)
        rk, rv = c.update_and_fetch(k, v)
        rows.append(dict(offset=int(c.offset), capacity=int(c.keys.shape[2]), keys=chk(rk), values=chk(rv)))
    t1 = int(c.trim(5))
    o1 = int(c.offset)
    t2 = int(c.trim(10 ** 6))
    out["kvcache"] = dict(rows=rows, trim5=t1, offset_after=o1, trim_all=t2, offset_end=int(c.offset))
    # merge -> step -> extract
    singles = []
    for i, n in enumerate((5, 2, 9)):
        s_ = C.KVCache()
        k, v = kv(1, n, 10 + i)
        s_.update_and_fetch(k, v)
        singles.append(s_)
    b = C.BatchKVCache.merge(singles)
    m = dict(left_padding=[int(x) for x in np.asarray(b.left_padding)], offset=[int(x) for x in np.asarray(b.offset)], size=int(b.size()))
    k, v = kv(3, 1, 99)
    keys, vals = b.update_and_fetch(k, v)
    m.update(step_keys=chk(keys), step_offset=[int(x) for x in np.asarray(b.offset)],
             extracted=[dict(offset=int(b.extract(i).offset), keys=chk(b.extract(i).state[0])) for i in range(3)])
    out["merge"] = m
    # filter / extend / trim
    b = C.BatchKVCache([1, 3, 0])
    k, v = kv(3, 4, 1)
    b.update_and_fetch(k, v)
    f = dict(offset0=[int(x) for x in np.asarray(b.offset)], idx0=int(b._idx))
    b.filter(mx.array([0, 1]))
    f.update(left_padding1=[int(x) for x in np.asarray(b.left_padding)], idx1=int(b._idx), keys1=chk(b.keys[..., : b._idx, :]))
    other = C.BatchKVCache([0])
    k2, v2 = kv(1, 6, 2)
    other.update_and_fetch(k2, v2)
    b.extend(other)
    f.update(idx2=int(b._idx), left_padding2=[int(x) for x in np.asarray(b.left_padding)], offset2=[int(x) for x in np.asarray(b.offset)],
             keys2=chk(b.keys[..., : b._idx, :]))
    f.update(trim=int(b.trim(2)), idx3=int(b._idx), offset3=[int(x) for x in np.asarray(b.offset)])
    e1, e2 = C.BatchKVCache([2]), C.BatchKVCache([0, 1])
    e1.extend(e2)
    f.update(empty_extend_left_padding=[int(x) for x in np.asarray(e1.left_padding)], empty=bool(e1.empty()))
    out["filter_extend"] = f
    with open(os.path.join(HERE, "ref_cache.json"), "w") as fh:
        json.dump(out, fh)
    return {k: (len(v) if isinstance(v, (list, dict)) else v) for k, v in out.items()}


def run_kitten_generate(R):
    """The reference's ``Model.generate`` of KittenTTS (kitten_tts.py:419-751: chunking, voice alias / speed prior, cross-fade between chunks, tail trim,
    fade-out, trailing silence, GenerationResult bookkeeping) with the network call replaced by a deterministic waveform (pt_layouts.fake_kitten_wave)
    and espeak by a stand-in phonemizer: what ``generate`` does with the audio it is handed."""
    import json

    import pt_layouts as PT

    from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS

    T = R["kitten"]
    cfg = dict(KS.tiny_config(), voice_aliases={"kiki": "expr-voice-2-f"}, speed_priors={"expr-voice-2-f": 0.8})
    model = T.Model(T.ModelConfig.from_dict(cfg))
    model.voices = {"expr-voice-2-f": np.zeros((40, 256), dtype=np.float32)}

    class Phonemizer:
        def phonemize(self, texts):
            return [t.lower() for t in texts]

    model._phonemizer = Phonemizer()
    orig = T.Model.__call__
    calls = []

    def fake(self, input_ids, ref_s, speed=1.0, return_output=False):
        n = int(input_ids.shape[-1])
        calls.append((n, float(speed)))
        return mx.array(PT.fake_kitten_wave(n, float(speed)))[None, :]

    T.Model.__call__ = fake
    out = []
    try:
        for case in PT.KITTEN_GENERATE_CASES:
            calls.clear()
            res = list(model.generate(case["text"], voice="kiki", clean_text=False, **case["kw"]))
            out.append(dict(calls=list(calls), results=[dict(samples=int(r.samples), segment_idx=int(r.segment_idx), token_count=int(r.token_count),
                                                             n=int(np.asarray(r.audio).shape[0]), sum=float(np.asarray(r.audio, dtype=np.float64).sum()),
                                                             sq=float((np.asarray(r.audio, dtype=np.float64) ** 2).sum()),
                                                             head=[float(x) for x in np.asarray(r.audio)[:3]], tail=[float(x) for x in np.asarray(r.audio)[-3:]])
                                                        for r in res]))
    finally:
        T.Model.__call__ = orig
    with open(os.path.join(HERE, "ref_kitten_generate.json"), "w") as f:
        json.dump(out, f)
    return [(len(c["calls"]), [r["n"] for r in c["results"]]) for c in out]


def run_whisper_host():
    """Host helpers of the reference's Whisper decode (decoding.py): ``DecodingTask._get_initial_tokens`` (:525-551: sot sequence, prefix and prompt
    truncation), ``get_suppress_tokens`` (:80-112), ``MaximumLikelihoodRanker.rank`` (:212-235) and ``compression_ratio`` (:15-17) on
    ``pt_layouts.WHISPER_HOST_CASES``."""
    import json

    import pt_layouts as PT

    wh, dec = import_whisper()
    codec = PT.WhisperCodec()

    class Tok(FakeWhisperTokenizer):
        def encode(self, text):
            return codec.encode(text)

    tok = Tok()
    C = PT.WHISPER_HOST_CASES
    out = dict(initial=[], suppress=[], rank=[], ratio=[])
    for kw in C["initial"]:
        kw = dict(kw)
        sample_len = kw.pop("sample_len", None)
        opts = dec.DecodingOptions(language="en", **kw)
        task = dec.DecodingTask.__new__(dec.DecodingTask)
        task.options, task.tokenizer, task.n_ctx = opts, tok, PT.WHISPER_HOST_N_CTX
        task.sample_len = sample_len or PT.WHISPER_HOST_N_CTX // 2
        task.sot_sequence = tok.sot_sequence_including_notimestamps if opts.without_timestamps else tok.sot_sequence
        out["initial"].append([int(t) for t in task._get_initial_tokens()])
    for sup in C["suppress"]:
        out["suppress"].append([int(t) for t in dec.get_suppress_tokens(tok, sup)])
    for r in C["rank"]:
        out["rank"].append(int(dec.MaximumLikelihoodRanker(r["length_penalty"]).rank([r["tokens"]], [r["sum_logprobs"]])[0]))
    for t in C["text"]:
        out["ratio"].append(float(dec.compression_ratio(t)))
    with open(os.path.join(HERE, "ref_whisper_host.json"), "w") as f:
        json.dump(out, f)
    return {k: len(v) for k, v in out.items()}


def run_whisper_generate():
    """The reference's ``Model.generate`` of Whisper (whisper.py:799-1320: 30 s windows, temperature fallback, no-speech skipping, segment cutting at
    consecutive timestamps, seek advance, prompt conditioning and its reset, clip timestamps) with ``_prepare_audio`` replaced by a ramp mel and ``decode``
    by a script of DecodingResults (pt_layouts.WHISPER_GENERATE_CASES): every decode call's window / temperature / prompt, and the resulting segments."""
    import json

    import pt_layouts as PT

    from mlx_audio_amd.stt.models.whisper import synthetic as WS

    wh, dec = import_whisper()
    u = sys.modules["mlx_audio.utils"]
    for n in ("base_load_model", "get_model_path", "load_config"):
        setattr(u, n, None)
    su = _load("mlx_audio.stt.utils", f"{REF}/stt/utils.py")
    dims = WS.tiny_dims()
    rd = wh.ModelDimensions(n_mels=80, n_audio_ctx=1500, n_audio_state=64, n_audio_head=2, n_audio_layer=1, n_vocab=51865, n_text_ctx=448, n_text_state=64,
                            n_text_head=2, n_text_layer=1)
    codec = PT.WhisperCodec()

    class Tok(FakeWhisperTokenizer):
        def encode(self, text):
            return codec.encode(text)

        def decode(self, tokens):  # as HFTokenizerWrapper.decode (whisper.py:74-82): timestamp tokens are dropped
            return codec.decode([t for t in tokens if t < self.timestamp_begin])

    out = []
    for case in PT.WHISPER_GENERATE_CASES:
        model = wh.Model(rd, dtype=mx.float32)
        model.get_tokenizer = lambda language=None, task="transcribe": Tok()
        n = case["frames"] + 3000
        mel = mx.array(np.broadcast_to(np.arange(1, n + 1, dtype=np.float32)[:, None], (n, 80)).copy())
        model._prepare_audio = lambda audio, padding=0, mel=mel, case=case: (mel, case["frames"])
        script = list(case["script"])
        calls = []

        def decode(segment, options, script=script, calls=calls):
            spec = script.pop(0)
            if "tokens" not in spec:           # {temperature: spec}: stay on this entry until a temperature is accepted
                table = spec
                key = min(table, key=lambda t: abs(float(t) - float(options.temperature)))
                spec = table[key]
                if float(key) != max(float(t) for t in table):
                    script.insert(0, table)
            col = np.asarray(segment)[:, 0]
            calls.append(dict(first=float(col[0]), nonzero=int((col != 0).sum()), last_nonzero=float(col[col != 0][-1]) if (col != 0).any() else 0.0,
                              temperature=float(options.temperature), prompt=[int(t) for t in (options.prompt or [])]))
            return dec.DecodingResult(audio_features=None, language="en", tokens=list(spec["tokens"]), text=codec.decode(spec["tokens"]),
                                      avg_logprob=spec.get("avg_logprob", -0.1), no_speech_prob=spec.get("no_speech_prob", 0.0),
                                      temperature=float(options.temperature), compression_ratio=spec.get("compression_ratio", 1.0))

        model.decode = decode
        res = model.generate(np.zeros(16000, np.float32), language="en", **case["kw"])
        segs = [dict(id=s["id"], seek=int(s["seek"]), start=float(s["start"]), end=float(s["end"]), tokens=[int(t) for t in s["tokens"]], text=s["text"],
                     temperature=float(s["temperature"])) for s in res.segments]
        out.append(dict(name=case["name"], calls=calls, segments=segs, text=res.text, unused_script=len(script)))
    with open(os.path.join(HERE, "ref_whisper_generate.json"), "w") as f:
        json.dump(out, f)
    return [(c["name"], len(c["calls"]), len(c["segments"])) for c in out]


def main():
    R = import_reference()
    if "codec_encode" in sys.argv[1:]:   # only the codec ENCODE fixtures (round 5)
        efx = run_dac_encode(seed_w=31, seed_audio=4, n_samples=320 * 24 + 77)
        np.savez_compressed(os.path.join(HERE, "ref_dac_encode.npz"), **efx)
        print("dac encode:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in efx.items() if a != "config"}, float(efx["commitment_loss"]))
        cfx = run_dac_compress(seed_w=31, seed_audio=8)
        np.savez_compressed(os.path.join(HERE, "ref_dac_compress.npz"), **cfx)
        print("dac compress:", {a: (v.shape if hasattr(v, "shape") and v.shape else v) for a, v in cfx.items() if a != "config"})
        for dwise in (True, False):
            sfx = run_snac_encode(seed_w=33, seed_audio=6, n_samples=48 * 4 * 9 + 101, depthwise=dwise)
            np.savez_compressed(os.path.join(HERE, f"ref_snac_encode_{'dw' if dwise else 'dense'}.npz"), **sfx)
            print("snac encode:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in sfx.items() if a != "config"})
        print("encodec encode (mono):", run_encodec_encode(41, 3, ENCODEC_TINY, 16 * 61 + 5, "mono"))
        print("encodec encode (stereo):", run_encodec_encode(43, 5, ENCODEC_ENC_STEREO, 2600, "stereo"))
        return
    if "qwen3_clone" in sys.argv[1:]:   # only the voice-cloning fixtures (round 3)
        sfx = run_qwen3_speaker_encoder(seed_w=13, seed_mel=5, frames=37)
        np.savez_compressed(os.path.join(HERE, "ref_qwen3_speaker_encoder.npz"), **sfx)
        print("qwen3 speaker encoder:", sfx["embedding"].shape, "peak", float(np.abs(sfx["embedding"]).max()))
        ifx = run_qwen3_icl(seed=19)
        np.savez_compressed(os.path.join(HERE, "ref_qwen3_icl.npz"), **ifx)
        print("qwen3 icl:", {a: v.shape for a, v in ifx.items() if hasattr(v, "shape") and a.startswith(("icl_embeds", "batch_embeds", "xvec_embeds", "decoded"))})
        print(ifx["tables"])
        return
    n = check_shim_against_reference_vectors(R)
    print(f"stand-in passes the reference's ConvTranspose / MLXSTFT vectors (both model families) and {n} interpolate vectors")
    k = run_kokoro(R, seed_w=21, n_phon=10, seed_ids=4, speed=1.0, seed_rng=5)
    np.savez_compressed(os.path.join(HERE, "ref_kokoro_tiny.npz"), **k)
    k32 = run_kokoro(R, seed_w=23, n_phon=8, seed_ids=7, speed=1.0, seed_rng=6, off_grid=True)
    np.savez_compressed(os.path.join(HERE, "ref_kokoro_tiny_f32.npz"), **{a: v for a, v in k32.items() if a not in ("xg_every8", "har_src")})
    print("kokoro:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in k.items()})
    for quant in (False, True):
        t = run_kitten(R, seed_w=22, n_phon=9, seed_ids=6, speed=1.1, seed_rng=8, quant=quant)
        np.savez_compressed(os.path.join(HERE, f"ref_kitten_tiny_{'quant' if quant else 'plain'}.npz"), **t)
        print("kitten quant" if quant else "kitten plain", {a: (v.shape if hasattr(v, "shape") else v) for a, v in t.items() if a not in ("flagged_modules", "all_modules")},
              len(t["flagged_modules"]), "flagged modules")
    mfx = run_mimi(seed_w=5, seed_codes=1, n_frames=30)
    np.savez_compressed(os.path.join(HERE, "ref_mimi_tiny.npz"), **mfx)
    print("mimi:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in mfx.items()},
          "decode vs decode_step max diff", float(np.abs(mfx["pcm"] - mfx["pcm_steps"]).max()), "peak", float(np.abs(mfx["pcm"]).max()))
    mefx = run_mimi_encode(seed_w=5, seed_audio=2, n_samples=1920 * 9 + 700)
    np.savez_compressed(os.path.join(HERE, "ref_mimi_encode.npz"), **mefx)
    print("mimi encode:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in mefx.items()})
    qefx = run_qwen3_tokenizer_encode(seed_w=9, seed_audio=3, n_samples=1920 * 8 + 1001)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_tokenizer_encode.npz"), **qefx)
    print("qwen3 tokenizer encode:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in qefx.items()})
    qfx = run_qwen3_talker(seed_w=1, seed_in=4)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_talker_tiny.npz"), **qfx)
    print("qwen3 talker:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in qfx.items()})
    cfx = run_qwen3_codec(seed_w=2, seed_codes=3, n_frames=40)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_codec_tiny.npz"), **cfx)
    print("qwen3 codec:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in cfx.items() if a != "missing"}, "missing", cfx["missing"].tolist()[:6],
          "peak", float(np.abs(cfx["audio"]).max()))
    sfx = run_csm(seed_w=2, seed_in=6, n_frames=3)
    np.savez_compressed(os.path.join(HERE, "ref_csm_tiny.npz"), **sfx)
    print("csm:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in sfx.items()})
    import json

    dfx = run_dac(seed_w=11, seed_codes=5, n_frames=37)
    np.savez_compressed(os.path.join(HERE, "ref_dac_tiny.npz"), **dfx)
    print("dac:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in dfx.items()}, "peak", float(np.abs(dfx["audio"]).max()))
    nfx, ncfg = run_snac(seed_w=4, seed_codes=9, n_frames=24)
    np.savez_compressed(os.path.join(HERE, "ref_snac_tiny.npz"), config=json.dumps(ncfg), **nfx)
    print("snac:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in nfx.items()}, "peak", float(np.abs(nfx["audio"]).max()))
    print("snac (LocalMHA) probe:", run_snac_local_mha_probe())
    vfx, vcfg = run_vocos(seed_w=3, seed_audio=1)
    np.savez_compressed(os.path.join(HERE, "ref_vocos_tiny.npz"), config=json.dumps(vcfg), **vfx)
    print("vocos:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in vfx.items()}, "peak", float(np.abs(vfx["audio"]).max()))
    pfx = run_sampler(seed=12)
    np.savez_compressed(os.path.join(HERE, "ref_sampler.npz"), **pfx)
    print("sampler:", sorted(pfx.keys()))
    xfx = run_dsp(R, seed=21)
    np.savez_compressed(os.path.join(HERE, "ref_dsp.npz"), **xfx)
    print("dsp:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in xfx.items()})
    print("sanitize:", run_sanitize(R))
    print("cache:", run_cache())
    print("kitten generate:", run_kitten_generate(R))
    bfx = run_bigvgan(seed_w=6, seed_mel=2, n_frames=50)
    np.savez_compressed(os.path.join(HERE, "ref_bigvgan_tiny.npz"), config=json.dumps(BIGVGAN_TINY), **bfx)
    print("bigvgan:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in bfx.items()}, "peak", float(np.abs(bfx["audio1"]).max()))
    print("dataclasses:", run_dataclasses(R))
    print("encodec:", run_encodec(3, 5, 23))
    print("broker:", run_broker())
    print("qwen3 session:", run_qwen3_session())
    print("whisper host helpers:", run_whisper_host())
    print("kokoro pipeline:", run_kokoro_pipeline())
    print("csm generate:", run_csm_generate())
    lfx = run_qwen3_generate_loop(seed_w=7)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_generate_loop.npz"), **lfx)
    print("qwen3 generate loop:", lfx["budget_codes"][:, 0].tolist(), "eos", lfx["eos_id"], lfx["eos_codes"].shape)
    sfx = run_qwen3_speaker_encoder(seed_w=13, seed_mel=5, frames=37)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_speaker_encoder.npz"), **sfx)
    ifx = run_qwen3_icl(seed=19)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_icl.npz"), **ifx)
    print("qwen3 clone:", sfx["embedding"].shape, len(ifx))
    qfx = run_qwen3_inputs(seed=17)
    np.savez_compressed(os.path.join(HERE, "ref_qwen3_inputs.npz"), **qfx)
    print("qwen3 inputs:", {a: v.shape for a, v in qfx.items() if hasattr(v, "shape") and a.startswith(("embeds", "batch"))})
    wfx = run_whisper(seed_w=3, seed_mel=2, sample_len=24)
    print("whisper generate:", run_whisper_generate())
    np.savez_compressed(os.path.join(HERE, "ref_whisper_tiny.npz"), **wfx)
    print("whisper:", {a: (v.shape if hasattr(v, "shape") else v) for a, v in wfx.items()}, wfx["ts_tokens"].tolist(), wfx["nots_tokens"].tolist())


if __name__ == "__main__":
    main()
