"""KittenTTS host logic and oracle pins that need no GPU: the quantiser's known answers, the reference's own KittenTTS tests
(tts/tests/test_models.py:416-492: config construction, ``sanitize`` of the dot-form Snake names, the phonemizer install hint) and the text helpers."""
import importlib
from unittest.mock import patch

import numpy as np
import pytest
import torch


def _config():
    """The config of the reference's TestKittenTTSModel (tts/tests/test_models.py:418-458)."""
    return {
        "hidden_dim": 16, "max_conv_dim": 16, "max_dur": 10, "n_layer": 1, "n_mels": 80, "n_token": 32, "style_dim": 64,
        "text_encoder_kernel_size": 3, "asr_res_dim": 8, "decoder_out_dim": 16,
        "plbert": {"num_hidden_layers": 1, "num_attention_heads": 1, "hidden_size": 16, "intermediate_size": 32, "max_position_embeddings": 32,
                   "embedding_size": 16, "inner_group_num": 1, "num_hidden_groups": 1, "hidden_dropout_prob": 0.0,
                   "attention_probs_dropout_prob": 0.0, "type_vocab_size": 2, "layer_norm_eps": 1e-12},
        "istftnet": {"resblock_kernel_sizes": [3, 3], "upsample_rates": [2, 2], "upsample_initial_channel": 32,
                     "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5]], "upsample_kernel_sizes": [4, 4], "gen_istft_n_fft": 16,
                     "gen_istft_hop_size": 4},
    }


def test_fake_quant_known_answers():
    """tts/models/kitten_tts/quant.py:4-20 by hand.  Range [-100, 155]: scale exactly 1, zero point 100, grid positions 100.5 / 101.5 / 102.5 round
    half to EVEN (100 / 102 / 102).  Range [-1, 1]: float32(2/255) is a hair above 2/255, so 1/scale = 127.49999 -> zero point 127 (not 128)."""
    from oracle.kokoro_ref import fake_quant_dynamic_u8

    got = fake_quant_dynamic_u8(torch.tensor([-100.0, 0.5, 1.5, 2.5, 155.0])).numpy()
    assert np.array_equal(got, np.array([-100, 0, 2, 2, 155], dtype=np.float32))

    s = np.float32(2.0) / np.float32(255.0)
    got = fake_quant_dynamic_u8(torch.tensor([-1.0, 0.0, 0.5, 1.0])).numpy()
    assert np.array_equal(got, np.array([-127, 0, 64, 127], dtype=np.float32) * s)
    # an all-positive tensor: the range is joined with 0, zero point 0, the grid reproduces multiples of max/255
    s3 = np.float32(3.0) / np.float32(255.0)
    got = fake_quant_dynamic_u8(torch.tensor([1.0, 2.0, 3.0])).numpy()
    assert np.array_equal(got, np.array([85, 170, 255], dtype=np.float32) * s3)
    assert float(fake_quant_dynamic_u8(torch.zeros(5)).abs().max()) == 0.0
    assert fake_quant_dynamic_u8(torch.randn(7).double()).dtype == torch.float64


def test_quant_flag_rule():
    """kitten_tts.py:291-299: a module is flagged when a listed name is the module or lies below it."""
    from oracle.kokoro_ref import P

    p = P({}, "", quant_modules=("decoder.encode.norm1.fc", "predictor.lstm"))
    assert p.sub("decoder").quant and p.sub("decoder.encode.norm1").quant and p.sub("decoder.encode.norm1.fc").quant
    assert not p.sub("decoder.encode.norm2").quant and not p.sub("decoder.encode.norm1.fc.weight").quant
    assert p.sub("predictor.lstm").quant and not p.sub("predictor.lstms").quant and not P({}, "").sub("decoder").quant


def test_model_config_and_init():
    from mlx_audio_amd.tts.models.kitten_tts import Model, ModelConfig

    cfg = _config()
    model = Model(ModelConfig.from_dict({**cfg, "not_a_field": 1}))
    assert model.config.n_token == cfg["n_token"] and model.config.voices_path == "voices.npz" and model.sample_rate == 24000
    assert model.config.decoder_out_dim == 16 and model.config.activation_quant_modules is None and model.engine is None
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, dtype=torch.int32), torch.zeros(1, 256))


def test_sanitize_alpha_names():
    from mlx_audio_amd.tts.models.kitten_tts import Model, ModelConfig

    model = Model(ModelConfig.from_dict(_config()))
    weights = {"decoder.generator.resblocks.0.alpha1.0": torch.ones(1, 1, 1), "decoder.generator.resblocks.0.alpha2.0": torch.ones(1, 1, 1)}
    out = model.sanitize(weights)
    assert "decoder.generator.resblocks.0.alpha1_0" in out and "decoder.generator.resblocks.0.alpha2_0" in out
    assert "decoder.generator.resblocks.0.alpha1.0" not in out
    new = {"decoder.generator.resblocks.0.alpha1_0": torch.ones(1, 1, 1)}
    assert model.sanitize(new) is new


def test_missing_phonemizer_error():
    from mlx_audio_amd.tts.models.kitten_tts import Model, ModelConfig

    model = Model(ModelConfig.from_dict(_config()))
    with patch("mlx_audio_amd.tts.models.kitten_tts.kitten_tts.importlib.import_module", side_effect=ModuleNotFoundError("No module named 'phonemizer'")):
        with pytest.raises(ImportError, match="pip install phonemizer-fork"):
            model._get_phonemizer()


def test_text_helpers():
    from mlx_audio_amd.tts.models.kitten_tts.kitten_tts import TextCleaner, basic_english_tokenize, chunk_text, ensure_punctuation

    assert basic_english_tokenize("hɛˈloʊ, wɜːld!") == ["hɛˈloʊ", ",", "wɜːld", "!"]  # stress marks count as word characters
    assert ensure_punctuation("  hi ") == "hi," and ensure_punctuation("hi!") == "hi!" and ensure_punctuation("") == ""
    assert chunk_text("One. Two three? four", max_len=400) == ["One,", "Two three,", "four,"]
    assert chunk_text("aaa bbb ccc ddd", max_len=7) == ["aaa bbb,", "ccc ddd,"]
    tc = TextCleaner()
    assert tc("$") == [0] and tc("A") == [17] and tc("a b") == [43, 16, 44] and tc("中") == []
    assert tc('"') == [15]  # the symbol list repeats the double quote; the LAST position wins, as in the reference's dict comprehension


def test_registry_knows_kitten():
    from mlx_audio_amd import registry

    assert registry.classify_model("kitten_tts", "") == "tts"
    mod = importlib.import_module("mlx_audio_amd.tts.models.kitten_tts")
    assert hasattr(mod, "Model") and hasattr(mod, "ModelConfig")


def test_oracle_without_quantisation_is_kokoro_with_other_widths():
    """With an empty module list and Kokoro's own widths / exact-GELU swapped in, the KittenTTS oracle path is the Kokoro oracle path: pins the
    shared plumbing (durations, alignment, decoder wiring) of oracle/kitten_ref.py to oracle/kokoro_ref.py (which tests/test_reference_fixtures_cpu.py
    pins to the reference's own modules, as it does oracle/kitten_ref.py directly)."""
    from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from oracle import kitten_ref
    from oracle.kokoro_ref import KokoroRef

    cfg = KS.tiny_config()
    w = KS.make_kitten_weights(cfg, seed=2)
    ids = S.make_phoneme_ids(8, seed=1)
    ref_s = S.make_voice_pack()[4]
    kit = kitten_ref.KittenRef(w, cfg, param_dtype=torch.bfloat16)
    kok = KokoroRef(w, cfg, param_dtype=torch.bfloat16)
    with patch.object(kitten_ref, "gelu_tanh_onnx", torch.nn.functional.gelu):
        pd, d, raw = kit.durations(ids, ref_s)
        pk, dk, rawk = kok.durations(ids, ref_s)
        assert torch.equal(pd, pk) and torch.equal(d, dk)
        rng = np.random.default_rng(0)
        F = int(pd.sum())
        ri, nz = rng.uniform(size=(1, 9)).astype(np.float32), rng.standard_normal((1, 2 * F * 300, 9)).astype(np.float32)
        a, _, ta = kit.forward(ids, ref_s, rand_ini=ri, noise=nz, return_intermediates=True)
        b, _, tb = kok.forward(ids, ref_s, rand_ini=ri, noise=nz, return_intermediates=True)
    assert all(torch.equal(ta[k], tb[k]) for k in ("f0", "n", "asr", "xg"))
    # ... up to the harmonic source: KittenTTS's coarse phase grid has 2F + 1 points (kitten_tts/istftnet.py:572), Kokoro's 2F
    assert a.shape == b.shape and not torch.equal(ta["har_src"], tb["har_src"])
    # and the tanh GELU / the quantiser do change the result
    assert not torch.equal(kit.durations(ids, ref_s)[1], dk)
    q = kitten_ref.KittenRef(w, dict(cfg, activation_quant_modules=KS.converter_quant_modules(w)), param_dtype=torch.bfloat16)
    assert not torch.equal(q.durations(ids, ref_s)[1], kit.durations(ids, ref_s)[1])


def test_generate_host_logic_matches_the_reference_generate():
    """tests/golden/ref_kitten_generate.json = the reference's own ``Model.generate`` (kitten_tts.py:419-751, run by tests/golden/make_reference_fixtures.py
    over the numpy stand-in for MLX) with the network call replaced by a deterministic waveform and espeak by a stand-in: chunking, alias and compounding
    speed prior (the (tokens, speed) of every network call), cross-fade, tail trim (the waveforms end in silence + a spurt), fade-out, trailing silence,
    segment / token bookkeeping.  This package's ``generate`` -- same stand-ins -- yields the same audio, sample for sample."""
    import json
    import os
    import sys

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import pt_layouts as PT

    from mlx_audio_amd.tts.models.kitten_tts import Model, ModelConfig
    from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS

    want = json.load(open(os.path.join(gold, "ref_kitten_generate.json")))
    calls = []

    class Probe(Model):
        def __call__(self, input_ids, ref_s, speed=1.0, return_output=False):
            n = int(input_ids.shape[-1])
            calls.append([n, float(speed)])
            return torch.from_numpy(PT.fake_kitten_wave(n, float(speed)))[None, :]

    class Phonemizer:
        def phonemize(self, texts):
            return [t.lower() for t in texts]

    cfg = dict(KS.tiny_config(), voice_aliases={"kiki": "expr-voice-2-f"}, speed_priors={"expr-voice-2-f": 0.8})
    model = Probe(ModelConfig.from_dict(cfg))
    model.voices = {"expr-voice-2-f": np.zeros((40, 256), dtype=np.float32)}
    model._phonemizer = Phonemizer()
    # default arguments (clean_text=True) with no text_preprocessor installed: works out of the box (one warning), same result as clean_text=False
    import warnings

    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter("always")
        res_default = list(model.generate(PT.KITTEN_GENERATE_CASES[0]["text"], voice="kiki"))
        list(model.generate(PT.KITTEN_GENERATE_CASES[0]["text"], voice="kiki"))
    assert len([w for w in wlog if "text_preprocessor" in str(w.message)]) == 1
    calls_default = [list(c) for c in calls]
    calls.clear()
    res_plain = list(model.generate(PT.KITTEN_GENERATE_CASES[0]["text"], voice="kiki", clean_text=False))
    assert calls_default[: len(calls)] == calls and len(res_default) == len(res_plain) and torch.equal(res_default[0].audio, res_plain[0].audio)
    model.text_preprocessor = lambda t: t.upper()   # an installed preprocessor is applied
    calls.clear()
    list(model.generate("ab", voice="kiki"))
    model.text_preprocessor = None
    for case, exp in zip(PT.KITTEN_GENERATE_CASES, want):
        calls.clear()
        res = list(model.generate(case["text"], voice="kiki", clean_text=False, **case["kw"]))
        assert [[c[0], round(c[1], 6)] for c in calls] == [[c[0], round(c[1], 6)] for c in exp["calls"]], (calls, exp["calls"])
        assert len(res) == len(exp["results"])
        for r, e in zip(res, exp["results"]):
            a = r.audio.double().numpy()
            assert r.samples == e["samples"] == a.shape[0] == e["n"] and r.segment_idx == e["segment_idx"] and r.token_count == e["token_count"]
            assert abs(a.sum() - e["sum"]) <= 1e-4 * (1 + abs(e["sum"])) and abs((a ** 2).sum() - e["sq"]) <= 1e-5 * (1 + e["sq"])
            assert np.allclose(a[:3], e["head"], atol=1e-6) and np.allclose(a[-3:], e["tail"], atol=1e-6)
