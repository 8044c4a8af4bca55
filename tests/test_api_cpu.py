"""Host-side mirror of the reference's Python surface for the hot path, CPU-only parts:
registry, loader error behaviour, Kokoro config / sanitize / pipeline plumbing, dsp constants and import isolation."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import dsp_ref


def test_registry_is_import_free_and_classifies():
    code = ("import sys; import mlx_audio_amd.registry as r; "
            "assert 'torch' not in sys.modules and 'mlx_audio_amd.ops' not in sys.modules; "
            "print(r.kinds(), r.classify_model('kokoro'), r.classify_model('', 'prince-canuma/Kokoro-82M'), r.classify_model('llama'), "
            "r.is_supported_model('whisper'), r.classify_model('whisper'), r.is_supported_model('parakeet'), r.classify_model('qwen3_tts'), "
            "r.classify_model('sesame'), r.classify_model('mimi'))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    # only loadable things are advertised: a kind needs its utils.py (codec ships decoder engines, no loader), a family its Model
    assert out.strip() == "('tts', 'stt') tts tts None True stt False tts tts None"
    from mlx_audio_amd import registry

    assert "kokoro" in registry.SUPPORTED_MODEL_TYPES["tts"]
    assert {"qwen3_tts", "sesame", "csm", "marvis"} <= registry.SUPPORTED_MODEL_TYPES["tts"]
    import importlib

    for kind, fams in registry.SUPPORTED_MODEL_TYPES.items():  # every advertised family really exposes Model (AST-level check, no import of torch)
        for fam in registry._families(kind):
            assert registry._exports_model(registry._PKG / kind / "models" / fam), (kind, fam)
    assert registry.supported_model_types("stt") == frozenset({"whisper"})


def test_dsp_imports_without_tts_or_stt():
    code = ("import sys; import mlx_audio_amd.dsp as d; "
            "bad = [m for m in sys.modules if m.startswith(('mlx_audio_amd.tts', 'mlx_audio_amd.stt', 'oracle'))]; assert not bad, bad; "
            "print(sorted(d.STR_TO_WINDOW_FN))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "['bartlett', 'blackman', 'hamming', 'hann', 'hanning']"


@pytest.mark.parametrize("name", ["hanning", "hamming", "blackman", "bartlett"])
@pytest.mark.parametrize("size,periodic", [(20, True), (400, False), (1024, False), (21, True)])
def test_dsp_windows_match_oracle(name, size, periodic):
    from mlx_audio_amd import dsp

    got = getattr(dsp, name)(size, periodic).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, getattr(dsp_ref, name)(size, periodic))


@pytest.mark.parametrize("args,kw", [((16000, 400, 80), dict(norm="slaney", mel_scale=None)),
                                     ((24000, 1024, 128, 0.0, 12000.0), dict(norm="slaney", mel_scale="slaney")),
                                     ((16000, 512, 40), dict(mel_scale="htk")),
                                     ((16000, 400, 80), dict(norm="slaney", mel_scale=None, precise=True))])
def test_dsp_mel_filters_match_oracle(args, kw):
    from mlx_audio_amd import dsp

    got = dsp.mel_filters(*args, **kw).numpy()
    want = dsp_ref.mel_filters(*args, **kw)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_dsp_argument_errors_match_reference():
    from mlx_audio_amd import dsp

    with pytest.raises(ValueError, match="Unknown window function"):
        dsp._resolve_window("kaiser", 16, False)
    assert dsp._resolve_window("hann", 20, True).shape[0] == 20
    assert np.array_equal(dsp._resolve_window("hann", 20, True).numpy(), dsp_ref.hanning(21)[:-1])  # periodic in istft


def test_kokoro_config_and_sanitize():
    from mlx_audio_amd.tts.models.kokoro import Model, ModelConfig
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    cfg = ModelConfig.from_dict({**S.KOKORO_CONFIG, "unknown_key": 1})
    assert cfg.sample_rate == 24000 and cfg.n_token == 178
    m = Model(cfg)
    assert m.sample_rate == 24000 and m.context_length == 512
    g = torch.Generator().manual_seed(0)
    raw = {
        "bert.embeddings.position_ids": torch.arange(4),
        "bert.pooler.weight": torch.randn(4, 4, generator=g),
        "bert_encoder.weight": torch.randn(4, 4, generator=g),
        "text_encoder.cnn.0.1.gamma": torch.ones(8),
        "text_encoder.cnn.0.1.beta": torch.zeros(8),
        "text_encoder.cnn.0.0.weight_v": torch.randn(8, 6, 5, generator=g),        # torch (out, in, K) -> transposed
        "text_encoder.lstm.weight_ih_l0_reverse": torch.randn(16, 8, generator=g),
        "predictor.lstm.bias_hh_l0": torch.randn(16, generator=g),
        "predictor.F0_proj.weight": torch.randn(1, 256, 1, generator=g),
        "predictor.F0.1.pool.weight_v": torch.randn(512, 1, 3, generator=g),         # depthwise (C, 1, 3) -> (C, 3, 1)
        "decoder.generator.noise_convs.0.weight": torch.randn(256, 22, 12, generator=g),
        "decoder.generator.resblocks.0.convs1.0.weight_v": torch.randn(256, 3, 3, generator=g),  # K == in: MLX layout already
        "decoder.generator.conv_post.weight_g": torch.randn(22, 1, 1, generator=g),
    }
    out = m.sanitize(raw)
    assert "bert.embeddings.position_ids" not in out
    assert "text_encoder.cnn.0.1.weight" in out and "text_encoder.cnn.0.1.bias" in out
    assert out["text_encoder.cnn.0.0.weight_v"].shape == (8, 5, 6)
    assert "text_encoder.lstm.Wx_backward" in out and "predictor.lstm.bias_hh_forward" in out
    assert out["predictor.F0_proj.weight"].shape == (1, 1, 256)
    assert out["predictor.F0.1.pool.weight_v"].shape == (512, 3, 1)
    assert out["decoder.generator.noise_convs.0.weight"].shape == (256, 12, 22)
    assert out["decoder.generator.resblocks.0.convs1.0.weight_v"].shape == (256, 3, 3)
    # key set is preserved on an already-converted checkpoint (layouts follow the reference's shape heuristic, which is
    # only meaningful on PyTorch-layout inputs: e.g. F0_proj.weight is always transposed, kokoro.py:231-243)
    w = S.make_kokoro_weights(S.tiny_config())
    assert set(m.sanitize(w)) == set(w)
    with pytest.raises(RuntimeError, match="no weights"):
        m("abc", torch.zeros(1, 256))
    ids = m.phonemes_to_ids("".join(list(cfg.vocab)[:5]) + "?")  # unknown symbols are dropped (kokoro.py:119-121)
    assert ids.tolist()[0] == 0 and ids.tolist()[-1] == 0 and len(ids) == 7


def test_pipeline_chunking_voices_and_errors(tmp_path):
    from mlx_audio_amd.tts.models.kokoro import KokoroPipeline

    p = KokoroPipeline("en-us", model=False, repo_id=str(tmp_path), g2p=lambda t: t.upper())
    assert p.lang_code == "a"
    res = list(p("hello\n\nworld"))
    assert [(r.graphemes, r.phonemes, r.audio) for r in res] == [("hello", "HELLO", None), ("world", "WORLD", None)]
    gs, ps, audio = res[0]
    assert (gs, ps, audio) == ("hello", "HELLO", None)
    chunks = KokoroPipeline.chunk_phonemes("ab. " * 300)
    assert all(len(c) <= 510 for c in chunks) and "".join(chunks).replace(" ", "") == ("ab." * 300)
    assert all(c.endswith(".") for c in chunks[:-1])  # sentence boundaries preferred
    with pytest.raises(AssertionError):
        KokoroPipeline("xx", model=False, repo_id="r")
    with pytest.raises(ValueError, match="repo_id"):
        KokoroPipeline("a", model=False, repo_id=None)
    # voices: blend of two packs = mean
    from safetensors.torch import save_file

    (tmp_path / "voices").mkdir()
    a, b = torch.randn(510, 1, 256), torch.randn(510, 1, 256)
    save_file({"voice": a}, str(tmp_path / "voices" / "va.safetensors"))
    torch.save(b, str(tmp_path / "voices" / "vb.pt"))
    assert torch.equal(p.load_voice("va"), a)
    assert torch.allclose(p.load_voice("va,vb"), (a + b) / 2)
    with pytest.raises(FileNotFoundError):
        p.load_voice("missing_voice")
    with pytest.raises(ValueError, match="Specify a voice"):
        list(KokoroPipeline("a", model=object(), repo_id=str(tmp_path), g2p=str)("hi"))
    with pytest.raises(ValueError, match="too long"):
        list(p.generate_from_tokens("a" * 511, voice=None))


def test_loader_error_behaviour(tmp_path):
    from mlx_audio_amd.tts.utils import load_model
    from mlx_audio_amd.utils import get_model_class, load_config, load_weights

    with pytest.raises(FileNotFoundError):
        load_model(tmp_path / "nope")
    with pytest.raises(FileNotFoundError):
        load_model("./definitely/not/here")
    d = tmp_path / "Kokoro-82M-bf16"
    d.mkdir()
    with pytest.raises(FileNotFoundError, match="Config not found"):
        load_config(d)
    (d / "config.json").write_text(json.dumps({"model_type": "does_not_exist"}))
    with pytest.raises(ValueError, match="not supported for tts"):
        get_model_class("does_not_exist", None, "tts", {})
    with pytest.raises(FileNotFoundError, match="No safetensors"):
        load_weights(d)
    with pytest.raises(ValueError, match="Invalid model path type"):
        load_model(123)
    arch, mt = get_model_class("styletts2", ["kokoro", "82m"], "tts", {"styletts2": "kokoro"})
    assert mt == "kokoro" and hasattr(arch, "Model") and hasattr(arch, "ModelConfig")


def test_generation_result_fields_match_reference():
    from mlx_audio_amd.tts.models.base import BatchGenerationResult, GenerationResult, check_array_shape, format_duration

    assert [f for f in GenerationResult.__dataclass_fields__] == [
        "audio", "samples", "sample_rate", "segment_idx", "token_count", "audio_duration", "real_time_factor", "prompt",
        "audio_samples", "processing_time_seconds", "peak_memory_usage", "is_streaming_chunk", "is_final_chunk"]
    assert "sequence_idx" in BatchGenerationResult.__dataclass_fields__
    assert format_duration(6.6) == "00:00:06.599" or format_duration(6.6) == "00:00:06.600"
    assert format_duration(3725.25) == "01:62:05.250"  # minutes are not wrapped in the reference either (kokoro.py:337-342)
    assert check_array_shape(torch.zeros(8, 3, 3)) and not check_array_shape(torch.zeros(8, 6, 5)) and not check_array_shape(torch.zeros(2, 3))


# ------------------------------------------------------------------------------------------------ Whisper host side
def test_whisper_tokenizer_ids_and_options():
    from mlx_audio_amd.stt.models.whisper.decoding import DecodingOptions, _verify_options, get_suppress_tokens
    from mlx_audio_amd.stt.models.whisper.tokenizer import get_tokenizer
    from oracle.whisper_ref import TokenizerSpec

    tk, spec = get_tokenizer(True, language="en", task="transcribe"), TokenizerSpec()
    for name in ("eot", "sot", "translate", "transcribe", "sot_lm", "sot_prev", "no_speech", "no_timestamps", "timestamp_begin"):
        assert getattr(tk, name) == getattr(spec, name), name
    assert tk.sot_sequence == spec.sot_sequence == (50258, 50259, 50359)
    assert tk.sot_sequence_including_notimestamps[-1] == 50363
    assert get_tokenizer(True, language="de", task="translate").sot_sequence == (50258, 50261, 50358)
    en = get_tokenizer(False)
    assert (en.eot, en.sot, en.sot_sequence) == (50256, 50257, (50257,))
    sup = get_suppress_tokens(tk, "-1")
    assert set(sup) == {tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm, tk.no_speech}
    assert get_suppress_tokens(tk, "5,7")[:2] == (5, 7)
    with pytest.raises(ValueError):
        _verify_options(DecodingOptions(beam_size=2, best_of=3))
    with pytest.raises(ValueError):
        _verify_options(DecodingOptions(temperature=0.0, best_of=3))
    with pytest.raises(ValueError):
        _verify_options(DecodingOptions(patience=1.0))
    with pytest.raises(ValueError):
        _verify_options(DecodingOptions(length_penalty=2.0))
    with pytest.raises(NotImplementedError):  # decoding.py:478-479
        _verify_options(DecodingOptions(beam_size=5))


def test_whisper_config_and_sanitize():
    from mlx_audio_amd.stt.models.whisper import Model, ModelDimensions

    d = ModelDimensions.from_dict({"d_model": 384, "encoder_layers": 4, "decoder_layers": 4, "encoder_attention_heads": 6,
                                   "decoder_attention_heads": 6, "num_mel_bins": 80, "vocab_size": 51865})
    assert (d.n_audio_state, d.n_text_state, d.n_audio_layer, d.n_audio_head, d.n_mels, d.n_vocab) == (384, 384, 4, 6, 80, 51865)
    d2 = ModelDimensions.from_dict({"n_mels": 80, "n_audio_state": 512, "model_type": "whisper", "unknown": 1})
    assert d2.n_audio_state == 512
    m = Model(d, device="cpu")
    hf = {"model.encoder.conv1.weight": torch.zeros(384, 80, 3), "model.encoder.embed_positions.weight": torch.zeros(1500, 384),
          "model.decoder.embed_positions.weight": torch.zeros(448, 384), "model.decoder.embed_tokens.weight": torch.zeros(10, 384),
          "model.decoder.layers.0.encoder_attn.k_proj.weight": torch.zeros(384, 384),
          "model.encoder.layers.1.self_attn.out_proj.bias": torch.zeros(384), "model.decoder.layers.2.fc1.weight": torch.zeros(1536, 384),
          "model.encoder.layer_norm.weight": torch.zeros(384), "model.decoder.layers.3.final_layer_norm.bias": torch.zeros(384)}
    s = m.sanitize(hf)
    assert set(s) == {"encoder.conv1.weight", "decoder.positional_embedding", "decoder.token_embedding.weight",
                      "decoder.blocks.0.cross_attn.key.weight", "encoder.blocks.1.attn.out.bias", "decoder.blocks.2.mlp1.weight",
                      "encoder.ln_post.weight", "decoder.blocks.3.mlp_ln.bias"}
    assert tuple(s["encoder.conv1.weight"].shape) == (384, 3, 80) and s["encoder.conv1.weight"].dtype == torch.float16
    assert m.is_multilingual and m.num_languages == 99
    with pytest.raises(RuntimeError):
        m.embed_audio(torch.zeros(3000, 80))


def test_whisper_synthetic_checkpoint_shapes():
    from mlx_audio_amd.stt.models.whisper import synthetic as WS

    dims = WS.tiny_dims()
    w = WS.make_whisper_weights(dims, seed=0)
    assert tuple(w["encoder.conv1.weight"].shape) == (128, 3, 80) and tuple(w["encoder.conv2.weight"].shape) == (128, 3, 128)
    assert "encoder.blocks.0.attn.key.bias" not in w and "decoder.blocks.1.cross_attn.key.bias" not in w  # K has no bias
    assert tuple(w["decoder.token_embedding.weight"].shape) == (51865, 128)
    assert all(torch.equal(v, v.to(torch.float16).to(torch.float32)) for v in w.values())  # fp16-representable


def test_kaldi_fbank_oracle_and_host_filterbank():
    """compute_fbank_kaldi restatement: the reference's own shape pins (sts/tests/test_mossformer2_se.py:134-146, 173-208: [58, 60] for
    24000 samples, 40 ms / 8 ms frames at 48 kHz), an independent float64 Kaldi statement of one frame, and the host filterbank mirror."""
    from mlx_audio_amd import dsp

    np.random.seed(42)
    audio = np.random.randn(24000).astype(np.float32)
    fb = dsp_ref.compute_fbank_kaldi(audio, sample_rate=48000, win_len=1920, win_inc=384, num_mels=60, dither=0.0)
    assert fb.shape == (58, 60) and np.isfinite(fb).all()
    # frame 7 by the textbook (Kaldi feature-window.cc order: DC removal, pre-emphasis with x[-1] := x[0]... the reference keeps x[0]), float64
    t = 7
    fr = audio[t * 384: t * 384 + 1920].astype(np.float64)
    fr = fr - fr.mean()
    fr = np.concatenate([fr[:1], fr[1:] - 0.97 * fr[:-1]])
    fr = fr * (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(1920) / 1919))
    P = np.abs(np.fft.rfft(fr, 2048)) ** 2
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(24000.0)
    d = (hi - lo) / 61
    want = np.empty(60)
    fm = mel(48000.0 / 2048 * np.arange(1024))
    for m in range(60):
        l, c, r = lo + m * d, lo + (m + 1) * d, lo + (m + 2) * d
        wgt = np.maximum(0.0, np.minimum((fm - l) / (c - l), (r - fm) / (r - c)))
        want[m] = np.log(max((P[:1024] * wgt).sum(), 1e-8))
    assert np.abs(fb[t] - want).max() < 2e-3
    # host mirror of get_mel_banks_kaldi == oracle, bit for bit; edge cases
    for args in [(60, 2048, 48000.0, 20.0, 0.0), (80, 512, 16000.0, 20.0, -400.0), (23, 256, 8000.0, 0.0, 0.0)]:
        b, c = dsp.get_mel_banks_kaldi(*args)
        bo, co = dsp_ref.get_mel_banks_kaldi(*args)
        assert np.array_equal(b.numpy(), bo) and np.array_equal(c.numpy(), co.astype(np.float32))
    assert dsp_ref.compute_fbank_kaldi(audio[:100], dither=0.0).shape == (0, 60)          # shorter than one window (snip_edges)
    assert dsp_ref.compute_fbank_kaldi(audio[:23900], dither=0.0, snip_edges=False).shape == (62, 60)
    with pytest.raises(ValueError, match="reflected edges"):   # 24000 % 384 >= 192: the reference's strided view leaves its padded buffer
        dsp_ref.compute_fbank_kaldi(audio, dither=0.0, snip_edges=False)
    n = np.random.default_rng(1).standard_normal((58, 1920)).astype(np.float32)
    assert np.abs(dsp_ref.compute_fbank_kaldi(audio, noise=n) - fb).max() > 1e-3        # dither=1.0 default perturbs


def test_vocos_oracle_reference_shape_pins_and_config():
    """codec/tests/test_vocos.py:60-73: mel model, 120 000 zeros -> (119552,); mel [1, 468, 100]; weight names / shapes of from_hparams."""
    import torch
    from mlx_audio_amd.codec.models.vocos.vocos import ISTFTHead, MelSpectrogramFeatures, VocosBackbone, make_vocos_weights
    from oracle import vocos_ref

    cfg = {"feature_extractor": {"class_path": "vocos.feature_extractors.MelSpectrogramFeatures",
                                 "init_args": {"sample_rate": 24000, "n_fft": 1024, "hop_length": 256, "n_mels": 100}},
           "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 64, "intermediate_dim": 192, "num_layers": 2}},
           "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 64, "n_fft": 1024, "hop_length": 256}}}
    w = make_vocos_weights(cfg, seed=0)
    assert w["backbone.embed.weight"].shape == (64, 7, 100) and w["backbone.convnext.1.dwconv.weight"].shape == (64, 7, 1)
    assert w["head.out.weight"].shape == (1026, 64) and torch.allclose(w["backbone.convnext.0.gamma"].mean(), torch.tensor(0.5), atol=0.1)
    ref = vocos_ref.VocosRef(w, cfg)
    mel = vocos_ref.log_mel_spectrogram(np.zeros(120_000, np.float32))
    assert mel.shape == (1, 468, 100) and np.allclose(mel, np.log(np.float32(1e-5)))
    assert ref(np.zeros(120_000, np.float32)).shape == (119552,)
    assert ref.decode(mel).shape == (119552,)
    with pytest.raises(ValueError, match="Padding must be"):
        MelSpectrogramFeatures(padding="valid")
    bb = VocosBackbone(**cfg["backbone"]["init_args"])
    assert bb.layer_scale_init_value == 0.5 and not bb.adanorm and ISTFTHead(64, 1024, 256, padding="same").hop_length == 256


def test_dac_oracle_reference_length_pins():
    """codec/tests/test_descript.py:41-42, 74-75, 107-108: the transposed convs' extra sample (groups passed as output_padding)."""
    import torch
    from mlx_audio_amd.codec.models.descript.dac import make_dac_weights
    from oracle.dac_ref import DACDecoderRef

    w = make_dac_weights(32, [8, 5, 4, 2], 16, 2, 32, 8, seed=0)
    ref = DACDecoderRef(w, [8, 5, 4, 2], 2)
    z = ref.from_codes(torch.randint(0, 32, (1, 2, 250), generator=torch.Generator().manual_seed(0)))
    assert tuple(z.shape) == (1, 16, 250)
    y = ref.decode(z)
    assert tuple(y.shape) == (1, 80_043, 1) and float(y.abs().max()) <= 1.0
    assert tuple(ref.decode(z[:, :, :375 - 250 + 125][:, :, :0 + 125]).shape) == (1, 125 * 320 + 43, 1)
    w2 = make_dac_weights(32, [8, 8, 4, 2], 16, 2, 32, 8, seed=0)
    assert tuple(DACDecoderRef(w2, [8, 8, 4, 2], 2).decode(torch.zeros(1, 16, 430)).shape) == (1, 220_235, 1)


class _FakeKokoroEngine:
    """Stands in for the device engine in the host-logic tests of the batch session: 'audio' = 10 samples per token id, valued with the
    voice row's first element, so batching, ordering, voice-row selection and speed routing are all observable without a GPU."""

    def __init__(self):
        self.calls = []

    def forward(self, ids, ref_s, speed=1.0):
        self.calls.append(dict(n=len(ids), speed=speed, lens=[int(i.numel()) for i in ids]))
        return [torch.full((10 * int(i.numel()),), float(ref_s[b, 0])) for b, i in enumerate(ids)], [None] * len(ids)


def _session_model(tmp_path, max_batch=4, stream=False):
    from mlx_audio_amd.tts.continuous import TTSBatchOptions
    from mlx_audio_amd.tts.models.kokoro import Model, ModelConfig
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from mlx_audio_amd.tts.models.kokoro.pipeline import KokoroPipeline

    m = Model(ModelConfig.from_dict(S.KOKORO_CONFIG), repo_id=str(tmp_path))
    m.engine = _FakeKokoroEngine()
    pipe = KokoroPipeline("a", model=m, repo_id=str(tmp_path), g2p=lambda t: t)   # texts are already phoneme strings
    pack = torch.arange(510, dtype=torch.float32).reshape(510, 1, 1).repeat(1, 1, 256)  # row r holds the value r
    pipe.voices["v"] = pack
    pipe.voices["w"] = pack + 1000.0
    m._pipelines["a"] = pipe
    return m, m.create_tts_batch_session(TTSBatchOptions(max_batch_size=max_batch, stream=stream))


def test_kokoro_batch_session_protocol(tmp_path):
    """tts/continuous.py protocol on the host side: slots, one chunk per active sequence per step, speed groups, completion events with the
    concatenated waveform, per-request errors, cancel, streaming events."""
    from mlx_audio_amd.tts import continuous as C
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    vocab = [p for p in S.KOKORO_CONFIG["vocab"] if p.strip() and p not in "!.?…:;,— "]
    short = "".join(vocab[:12])
    long_text = (" ".join("".join(vocab[i:i + 40]) for i in range(0, 160, 40)) + " ") * 8
    long_text = long_text[:1200].strip()  # > 510 symbols: three chunks
    m, s = _session_model(tmp_path, max_batch=3)
    assert m.supports_tts_batch(voice="v") and m.supports_tts_continuous_batch(speed=1.3) and not m.supports_tts_batch(instruct="x")
    assert not m.supports_tts_batch(ref_audio=object()) and not m.supports_tts_batch(pitch=1.2)
    assert s.idle and s.available_slots == 3
    s.add([C.TTSBatchItem(0, long_text, voice="v"), C.TTSBatchItem(1, short, voice="w"), C.TTSBatchItem(2, short, voice="nope")])
    assert not s.idle and s.available_slots == 0
    with pytest.raises(ValueError):
        s.add([C.TTSBatchItem(3, short, voice="v")])
    ev = s.step()
    # the unknown voice fails alone; the short request finishes in the first pass; the long one needs more steps
    by_id = {e.sequence_id: e for e in ev}
    assert set(by_id) == {1, 2} and isinstance(by_id[2].error, FileNotFoundError) and by_id[2].done
    assert by_id[1].done and by_id[1].error is None and by_id[1].samples == 10 * (len(short) + 2) == by_id[1].audio.numel()
    assert float(by_id[1].audio[0]) == 1000.0 + len(short) - 1          # voice row = chunk length - 1 (pipeline.py:303), pack "w"
    assert m.engine.calls[0]["n"] == 2 and s.available_slots == 2
    s.add([C.TTSBatchItem(7, short, voice="v", speed=1.5)])               # joins between steps; its speed group runs after the leader's
    ev2 = s.step()
    assert m.engine.calls[1] == dict(n=1, speed=1.0, lens=m.engine.calls[1]["lens"]) and ev2 == []
    done = {}
    for _ in range(6):
        for e in s.step():
            done[e.sequence_id] = e
        if s.idle:
            break
    assert s.idle and set(done) == {0, 7}
    n_chunks = done[0].metadata["chunks"]
    assert n_chunks == 3 and done[0].token_count == sum(len(c) for c in m._pipelines["a"].chunk_phonemes(long_text))
    assert done[0].samples == done[0].audio.numel() and done[0].is_final_chunk and done[0].sample_rate == 24000
    assert any(c["speed"] == 1.5 for c in m.engine.calls) and float(done[7].audio[0]) == len(short) - 1
    # cancel
    s.add([C.TTSBatchItem(9, long_text, voice="v")])
    s.step()
    s.cancel(9)
    assert s.idle and s.step() == []
    # streaming: one event per chunk, the last one final
    m2, st = _session_model(tmp_path, stream=True)
    st.add([C.TTSBatchItem(0, long_text, voice="v")])
    evs = []
    while not st.idle:
        evs += st.step()
    assert len(evs) == 3 and all(e.is_streaming_chunk for e in evs) and [e.is_final_chunk for e in evs] == [False, False, True] and evs[-1].done
    # batch_generate(texts, voices=names): the serving shell's entry point (server.py:519-545 probes the parameter names)
    import inspect

    assert {"texts", "voices"} <= set(inspect.signature(m2.batch_generate).parameters)
    res = list(m2.batch_generate([short, long_text], voices=["v", "w"]))
    assert sorted(r.sequence_idx for r in res) == [0, 1] and all(r.samples == r.audio.numel() for r in res)
    with pytest.raises(FileNotFoundError):
        list(m2.batch_generate([short], voices="missing"))


def test_codec_package_exports_match_reference_names():
    """mlx_audio/codec/__init__.py and codec/models/__init__.py: the built decoders resolve under the reference's names (lazily, without a GPU);
    the codecs outside the scope raise ImportError, unknown names AttributeError."""
    import mlx_audio_amd.codec as C
    import mlx_audio_amd.codec.models as CM
    from mlx_audio_amd.codec.models.descript import DAC
    from mlx_audio_amd.codec.models.snac import SNAC

    assert C.DAC is DAC and CM.SNAC is SNAC and C.Vocos.__name__ == "Vocos" and CM.Mimi.__name__ == "MimiDecoder"
    assert set(CM.__all__) == {"DAC", "SNAC", "Vocos", "Mimi", "Encodec"} and CM.Encodec.__name__ == "Encodec"
    with pytest.raises(ImportError):
        CM.EcapaTdnnBackbone
    with pytest.raises(AttributeError):
        CM.NoSuchCodec
    # constructing an engine without a ROCm device fails loudly (no CPU fallback)
    if not torch.cuda.is_available():
        from mlx_audio_amd import _lib

        with pytest.raises(_lib.Mi355Error):
            SNAC(attn_window_size=None)


def _drain(handle, timeout=10.0):
    """All chunks of a request up to and including 'done'."""
    import queue as _q

    out = []
    while True:
        try:
            c = handle.result_queue.get(timeout=timeout)
        except _q.Empty:
            raise AssertionError(f"request {handle.context.request_id} never finished: {[x.kind for x in out]}")
        out.append(c)
        if c.kind == "done":
            return out


def test_inference_broker_continuous_batching(tmp_path):
    """mlx_audio/server_inference.py surface over the Kokoro batch session: concurrent requests share engine passes, a failing request gets its own
    error chunk, a cancelled one is dropped, more requests than slots wait in the backlog, stop_and_join fails what is still active."""
    from mlx_audio_amd.server_inference import BaseModelExecutionAdapter, InferenceBroker, TTSExecutionAdapter
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    vocab = [p for p in S.KOKORO_CONFIG["vocab"] if p.strip() and p not in "!.?…:;,— "]
    short = "".join(vocab[:12])
    m, _ = _session_model(tmp_path)
    broker = InferenceBroker(idle_poll_s=0.01)
    try:
        with pytest.raises(ValueError):
            broker.submit(endpoint_kind="tts", model_name="kokoro", payload={"text": short})
        broker.register_adapter("tts", TTSExecutionAdapter({"kokoro": m}, max_batch_size=2))
        hs = [broker.submit(endpoint_kind="tts", model_name="kokoro", payload={"text": short[: 6 + i], "voice": "v"}) for i in range(5)]   # 5 requests, 2 slots
        bad = broker.submit(endpoint_kind="tts", model_name="kokoro", payload={"text": short, "voice": "nope"})
        assert hs[0].context.batch_key == ("tts", False, "a") and hs[0].context.endpoint_kind == "tts"
        for i, h in enumerate(hs):
            chunks = _drain(h)
            assert [c.kind for c in chunks] == ["data", "done"], [c.kind for c in chunks]
            ev = chunks[0].payload
            assert ev.done and ev.samples == 10 * (6 + i + 2) and float(ev.audio[0]) == 6 + i - 1      # fake engine: 10 samples per id, value = voice row
        chunks = _drain(bad)
        assert [c.kind for c in chunks] == ["error", "done"] and isinstance(chunks[0].error, FileNotFoundError)
        assert max(c["n"] for c in m.engine.calls) <= 2 and sum(c["n"] for c in m.engine.calls) == 5   # two slots: passes of <= 2 utterances, 5 in total
        # (how many passes actually carried 2 depends on how the submitting thread and the worker interleave: not asserted)
        # unknown model: the session cannot be created -> error + done on that request only
        chunks = _drain(broker.submit(endpoint_kind="tts", model_name="other", payload={"text": short}))
        assert [c.kind for c in chunks] == ["error", "done"] and isinstance(chunks[0].error, ValueError)
        # a cancelled request never reaches the engine
        n_before = len(m.engine.calls)
        h = broker.submit(endpoint_kind="tts", model_name="kokoro", payload={"text": short, "voice": "v"})
        h.cancel()
        time_limit = 50
        while time_limit and not broker._inbox.empty():
            import time as _t

            _t.sleep(0.01)
            time_limit -= 1
        # serial fallback for a model without the session hooks
        class Plain:
            def generate(self, text, **kw):
                yield ("seg", text, kw.get("voice"))

        class Serial(BaseModelExecutionAdapter):
            def run_serial(self, request):
                for r in Plain().generate(request.payload["text"], voice=request.payload.get("voice")):
                    request.emit_data(r)
                request.emit_done()

        broker.register_adapter("plain", Serial())
        chunks = _drain(broker.submit(endpoint_kind="plain", model_name="x", payload={"text": "abc", "voice": "w"}))
        assert [c.kind for c in chunks] == ["data", "done"] and chunks[0].payload == ("seg", "abc", "w")
        assert len(m.engine.calls) in (n_before, n_before + 1)    # the cancelled request was dropped before or right after admission, never synthesised twice
    finally:
        broker.stop_and_join(timeout=5.0)
    assert not broker._worker.is_alive()


def test_utils_reexports_dsp_entry_points():
    """mlx_audio/utils.py:31-40 + tests/test_dsp.py:30-38: the dsp functions are importable from ``utils`` (backward-compatible path), lazily."""
    import subprocess
    import sys

    from mlx_audio_amd import dsp
    from mlx_audio_amd.utils import STR_TO_WINDOW_FN, bartlett, blackman, hamming, hanning, istft, mel_filters, stft

    assert stft is dsp.stft and istft is dsp.istft and mel_filters is dsp.mel_filters and hanning is dsp.hanning
    assert hamming is dsp.hamming and blackman is dsp.blackman and bartlett is dsp.bartlett and STR_TO_WINDOW_FN is dsp.STR_TO_WINDOW_FN
    code = "import sys, mlx_audio_amd.utils; assert 'mlx_audio_amd.dsp' not in sys.modules; import mlx_audio_amd.utils as u; u.stft; assert 'mlx_audio_amd.dsp' in sys.modules"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with pytest.raises(AttributeError):
        import mlx_audio_amd.utils as u

        u.no_such_name


def test_adjust_speed_and_stt_resample_helpers():
    """tts/models/base.py:37-68 (linear-interpolation speed change, both end points kept) and stt/utils.py:100-103 (time-first resample)."""
    from mlx_audio_amd.stt.utils import resample_audio
    from mlx_audio_amd.tts.models.base import adjust_speed

    x = torch.arange(11, dtype=torch.float32)
    y = adjust_speed(x, 2.0)
    assert y.shape == (5,) and torch.allclose(y, torch.tensor([0.0, 2.5, 5.0, 7.5, 10.0]))
    z = adjust_speed(np.stack([np.arange(9.0), -np.arange(9.0)], axis=1), 0.5)
    assert z.shape == (18, 2) and float(z[0, 0]) == 0.0 and float(z[-1, 0]) == 8.0 and torch.allclose(z[:, 0], -z[:, 1])
    a = resample_audio(np.zeros((24000, 2), dtype=np.float32), 24000, 16000)
    assert a.shape == (16000, 2) and a.dtype == np.float32
