"""Host logic of ``whisper.Model.generate`` (reference ``stt/models/whisper/whisper.py:799-1320``) with the device calls stubbed: 30 s windows
hold ONLY their own frames (zero-padded in the log-mel domain), temperature fallback, prompt conditioning, clip timestamps, segment cutting.
No GPU: ``decode`` / ``_prepare_audio`` are replaced by recorders."""
import numpy as np
import pytest
import torch

from mlx_audio_amd.stt.models.whisper import Model, ModelDimensions
from mlx_audio_amd.stt.models.whisper.audio import N_FRAMES
from mlx_audio_amd.stt.models.whisper.decoding import DecodingOptions, DecodingResult, initial_tokens, rank_group
from mlx_audio_amd.stt.models.whisper.tokenizer import get_tokenizer


class Codec:
    def encode(self, text):
        return [100 + (ord(c) % 50) for c in text]

    def decode(self, toks):
        return "".join(chr(ord("a") + (t % 26)) for t in toks)


def dims():
    return ModelDimensions(n_mels=80, n_audio_ctx=1500, n_audio_state=64, n_audio_head=2, n_audio_layer=1, n_vocab=51865, n_text_ctx=448,
                           n_text_state=64, n_text_head=2, n_text_layer=1)


class Stub(Model):
    """Model whose mel is a ramp (frame index in every bin) and whose decode replays a script of DecodingResults."""

    def __init__(self, content_frames, script):
        super().__init__(dims(), device="cpu")
        self.codec = Codec()
        self.content_frames = content_frames
        self.script = list(script)
        self.calls = []

    def _prepare_audio(self, audio, padding=0):
        n = self.content_frames + N_FRAMES
        mel = torch.arange(1, n + 1, dtype=torch.float32)[:, None].expand(n, 80).clone()
        return mel, self.content_frames

    def decode(self, mel, options=DecodingOptions(), **kw):
        self.calls.append(dict(mel=mel.clone(), options=options))
        spec = self.script.pop(0)
        spec = spec(options) if callable(spec) else spec
        return DecodingResult(audio_features=None, language="en", tokens=spec["tokens"], text=self.codec.decode(spec["tokens"]),
                              avg_logprob=spec.get("avg_logprob", -0.1), no_speech_prob=spec.get("no_speech_prob", 0.0),
                              temperature=options.temperature, compression_ratio=spec.get("compression_ratio", 1.0))


def tok():
    return get_tokenizer(True, language="en", task="transcribe", codec=Codec())


def test_window_holds_only_its_own_frames_zero_padded():
    # the last window must be mel[seek:seek+segment_size] padded with 0.0, not the audio/padding frames that follow it
    t = tok()
    tb = t.timestamp_begin
    m = Stub(content_frames=4000, script=[dict(tokens=[tb, 200, 201, tb + 1500]), dict(tokens=[tb, 300, tb + 100])])
    out = m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0)
    assert len(m.calls) == 2
    first, last = m.calls[0]["mel"], m.calls[1]["mel"]
    assert first.shape == last.shape == (N_FRAMES, 80)
    np.testing.assert_array_equal(first[:, 0].numpy(), np.arange(1, N_FRAMES + 1, dtype=np.float32))
    np.testing.assert_array_equal(last[:1000, 0].numpy(), np.arange(3001, 4001, dtype=np.float32))  # seek 3000, segment_size 1000
    assert float(last[1000:].abs().max()) == 0.0
    assert [s["seek"] for s in out.segments] == [0, 3000]
    assert [s["id"] for s in out.segments] == [0, 1]
    assert out.segments[0]["start"] == 0.0 and out.segments[0]["end"] == 30.0
    assert out.segments[1]["start"] == 30.0 and out.segments[1]["end"] == pytest.approx(32.0)


def test_temperature_fallback_thresholds_and_best_of_handling():
    t = tok()
    tb = t.timestamp_begin

    def bad(o):
        assert o.best_of is None and o.temperature == 0.0  # best_of is dropped at T = 0 (whisper.py:969-971)
        return dict(tokens=[tb, 200, tb + 1500], compression_ratio=3.0)

    def low(o):
        assert o.best_of == 5 and o.temperature == pytest.approx(0.2)
        return dict(tokens=[tb, 200, tb + 1500], avg_logprob=-2.0)

    def ok(o):
        assert o.temperature == pytest.approx(0.4)
        return dict(tokens=[tb, 200, tb + 1500])

    m = Stub(content_frames=3000, script=[bad, low, ok])
    out = m.generate(np.zeros(16000, np.float32), language="en", best_of=5)
    assert len(m.calls) == 3 and out.segments[0]["temperature"] == pytest.approx(0.4)
    # silence cancels the fallback (whisper.py:986-990) and the window is skipped (whisper.py:1056-1069)
    m = Stub(content_frames=3000, script=[dict(tokens=[tb, 200, tb + 1500], avg_logprob=-2.0, no_speech_prob=0.9)])
    out = m.generate(np.zeros(16000, np.float32), language="en")
    assert len(m.calls) == 1 and out.segments == [] and out.text == ""
    # a scalar temperature means one attempt
    m = Stub(content_frames=3000, script=[dict(tokens=[tb, 200, tb + 1500], compression_ratio=9.0)])
    m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0)
    assert len(m.calls) == 1


def test_prompt_conditioning_initial_prompt_and_reset():
    t = tok()
    tb = t.timestamp_begin
    win = [tb, 200, 201, tb + 1500]
    m = Stub(content_frames=9000, script=[dict(tokens=win), dict(tokens=win), dict(tokens=win)])
    out = m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0, initial_prompt="hi")
    ip = Codec().encode(" hi")
    assert m.calls[0]["options"].prompt == ip
    assert m.calls[1]["options"].prompt == ip + win
    assert m.calls[2]["options"].prompt == ip + win + win
    assert out.text == Codec().decode([200, 201] * 3)  # the initial prompt is not part of the text (whisper.py:1318)
    m = Stub(content_frames=6000, script=[dict(tokens=win), dict(tokens=win)])
    m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0, condition_on_previous_text=False)
    assert m.calls[1]["options"].prompt == []
    # a window decoded above T = 0.5 resets the prompt too (whisper.py:1298-1300)
    m = Stub(content_frames=6000, script=[dict(tokens=win), dict(tokens=win)])
    m.generate(np.zeros(16000, np.float32), language="en", temperature=0.8)
    assert m.calls[1]["options"].prompt == []
    # hotwords are folded into the prompt (stt/utils.py:15-34)
    m = Stub(content_frames=3000, script=[dict(tokens=win)])
    m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0, initial_prompt="hi", hotwords=["Foo", " ", "Bar"])
    assert m.calls[0]["options"].prompt == Codec().encode(" hi\nFoo, Bar")


def test_consecutive_timestamps_cut_segments_and_advance_seek():
    t = tok()
    tb = t.timestamp_begin
    toks = [tb, 200, tb + 100, tb + 100, 201, tb + 250, tb + 250, 202]  # two closed segments and an open tail
    m = Stub(content_frames=3000, script=[dict(tokens=toks), dict(tokens=[tb, 203, tb + 1000])])
    out = m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0)
    assert [(s["start"], s["end"]) for s in out.segments[:2]] == [(0.0, 2.0), (2.0, 5.0)]
    assert m.calls[1]["mel"][0, 0] == 501.0  # seek advanced to the last closed timestamp: 250 tokens * 2 frames
    assert out.segments[2]["seek"] == 500
    # empty / instantaneous segments are cleared (whisper.py:1269-1277)
    m = Stub(content_frames=3000, script=[dict(tokens=[tb + 5, tb + 5, 200, tb + 1500])])
    out = m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0)
    assert out.segments[0]["text"] == "" and out.segments[0]["tokens"] == []


def test_clip_timestamps_and_unsupported_options():
    t = tok()
    tb = t.timestamp_begin
    m = Stub(content_frames=6000, script=[dict(tokens=[tb, 200, tb + 500])])
    out = m.generate(np.zeros(16000, np.float32), language="en", temperature=0.0, clip_timestamps="10,20")
    assert len(m.calls) == 1
    mel = m.calls[0]["mel"]
    assert mel[0, 0] == 1001.0 and mel[999, 0] == 2000.0 and float(mel[1000:].abs().max()) == 0.0
    assert out.segments[0]["start"] == 10.0
    for kw in (dict(word_timestamps=True), dict(hallucination_silence_threshold=1.0), dict(stream=True)):
        with pytest.raises(NotImplementedError):
            Stub(3000, []).generate(np.zeros(16000, np.float32), language="en", **kw)


def test_initial_tokens_and_group_ranking():
    t = tok()
    sot = list(t.sot_sequence)
    assert initial_tokens(t, DecodingOptions(), 448, 224) == sot
    assert initial_tokens(t, DecodingOptions(without_timestamps=True), 448, 224) == list(t.sot_sequence_including_notimestamps)
    long_prompt = list(range(1000, 1300))
    got = initial_tokens(t, DecodingOptions(prompt=long_prompt), 448, 224)
    assert got[0] == t.sot_prev and got[1:224] == long_prompt[-223:] and got[224:] == sot  # decoding.py:545-549
    got = initial_tokens(t, DecodingOptions(prefix=list(range(10)), sample_len=220), 448, 220)
    assert got == sot + list(range(10))[-4:]  # decoding.py:534-537
    assert initial_tokens(t, DecodingOptions(prompt="ab"), 448, 224)[1:3] == Codec().encode(" ab")[:2]
    # MaximumLikelihoodRanker (decoding.py:212-235)
    assert rank_group([[1, 2, 3, 4], [1, 2]], [-4.0, -3.0], None) == 0
    assert rank_group([[1, 2, 3, 4], [1, 2]], [-4.0, -3.0], 0.0) == 1


def test_generate_host_loop_matches_the_reference_generate():
    """tests/golden/ref_whisper_generate.json = the reference's own ``Model.generate`` (whisper.py:799-1320, executed over the numpy stand-in for MLX by
    tests/golden/make_reference_fixtures.py) with ``_prepare_audio`` replaced by a ramp mel and ``decode`` by scripts of DecodingResults: nine scenarios
    (plain windows, temperature fallback, no-speech skip, segment cutting at consecutive timestamps with and without a trailing pair, prompt conditioning and
    its resets, clip timestamps, no timestamps).  This package's ``generate`` -- same stubs -- issues the same decode calls (window contents, temperature,
    prompt tokens) and returns the same segments and text."""
    import json
    import os
    import sys

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import pt_layouts as PT

    want = json.load(open(os.path.join(gold, "ref_whisper_generate.json")))
    codec = PT.WhisperCodec()

    class Replay(Model):
        def __init__(self, case):
            super().__init__(dims(), device="cpu")
            self.codec = codec
            self.case = case
            self.script = list(case["script"])
            self.calls = []

        def _prepare_audio(self, audio, padding=0):
            n = self.case["frames"] + N_FRAMES
            return torch.arange(1, n + 1, dtype=torch.float32)[:, None].expand(n, 80).clone(), self.case["frames"]

        def decode(self, mel, options=DecodingOptions(), **kw):
            spec = self.script.pop(0)
            if "tokens" not in spec:
                table = spec
                key = min(table, key=lambda t: abs(float(t) - float(options.temperature)))
                spec = table[key]
                if float(key) != max(float(t) for t in table):
                    self.script.insert(0, table)
            col = mel[:, 0].numpy()
            nz = col[col != 0]
            self.calls.append(dict(first=float(col[0]), nonzero=int((col != 0).sum()), last_nonzero=float(nz[-1]) if nz.size else 0.0,
                                   temperature=float(options.temperature), prompt=[int(t) for t in (options.prompt or [])]))
            return DecodingResult(audio_features=None, language="en", tokens=list(spec["tokens"]), text=codec.decode(spec["tokens"]),
                                  avg_logprob=spec.get("avg_logprob", -0.1), no_speech_prob=spec.get("no_speech_prob", 0.0),
                                  temperature=float(options.temperature), compression_ratio=spec.get("compression_ratio", 1.0))

        def get_tokenizer(self, language=None, task="transcribe"):
            return get_tokenizer(True, language=language or "en", task=task, codec=codec)

    assert len(want) == len(PT.WHISPER_GENERATE_CASES)
    for case, exp in zip(PT.WHISPER_GENERATE_CASES, want):
        # json turned the temperature keys of the fallback tables into strings on the reference side only; the scripts here are the python objects
        m = Replay(case)
        out = m.generate(np.zeros(16000, np.float32), language="en", **case["kw"])
        assert len(m.calls) == len(exp["calls"]), (case["name"], len(m.calls), len(exp["calls"]))
        for a, b in zip(m.calls, exp["calls"]):
            assert a["first"] == b["first"] and a["nonzero"] == b["nonzero"] and a["last_nonzero"] == b["last_nonzero"], (case["name"], a, b)
            assert abs(a["temperature"] - b["temperature"]) < 1e-9 and a["prompt"] == b["prompt"], (case["name"], a, b)
        assert len(m.script) == exp["unused_script"], case["name"]
        got = [dict(id=s["id"], seek=int(s["seek"]), start=float(s["start"]), end=float(s["end"]), tokens=[int(t) for t in s["tokens"]], text=s["text"],
                    temperature=float(s["temperature"])) for s in out.segments]
        assert len(got) == len(exp["segments"]), (case["name"], got, exp["segments"])
        for g, e in zip(got, exp["segments"]):
            assert g["id"] == e["id"] and g["seek"] == e["seek"] and g["tokens"] == e["tokens"] and g["text"] == e["text"], (case["name"], g, e)
            assert abs(g["start"] - e["start"]) < 1e-9 and abs(g["end"] - e["end"]) < 1e-9 and abs(g["temperature"] - e["temperature"]) < 1e-9, (case["name"], g, e)
        assert out.text == exp["text"], (case["name"], out.text, exp["text"])


def test_generate_detects_the_language_from_the_probability_dict():
    """whisper.py:897-905: with ``language=None`` a multilingual model asks ``detect_language`` (-> tokens, {code: probability}) and decodes in the most
    probable language; the first 30 s window is what it is shown."""
    seen = {}
    tb = tok().timestamp_begin

    class M(Stub):
        def detect_language(self, mel, tokenizer=None):
            seen["shape"] = tuple(mel.shape)
            return torch.tensor(50261), {"en": 0.2, "de": 0.7, "fr": 0.1}

    m = M(3000, [dict(tokens=[tb, 100, tb + 1500])])
    out = m.generate(np.zeros(16000, np.float32), temperature=0.0)
    assert [c["options"].language for c in m.calls] == ["de"] and out.language == "de" and seen["shape"][0] == N_FRAMES


def test_decode_host_helpers_match_the_reference():
    """``ref_whisper_host.json`` = the reference's ``DecodingTask._get_initial_tokens`` (decoding.py:525-551), ``get_suppress_tokens`` (:80-112),
    ``MaximumLikelihoodRanker.rank`` (:212-235) and ``compression_ratio`` (:15-17) on ``pt_layouts.WHISPER_HOST_CASES``; this package's
    ``initial_tokens`` / ``get_suppress_tokens`` / ``rank_group`` / ``compression_ratio`` give the same."""
    import json
    import os
    import sys

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import pt_layouts as PT

    from mlx_audio_amd.stt.models.whisper import decoding as D

    want = json.load(open(os.path.join(gold, "ref_whisper_host.json")))
    C = PT.WHISPER_HOST_CASES
    t = get_tokenizer(True, language="en", task="transcribe", codec=PT.WhisperCodec())
    t.non_speech_tokens = (1, 2, 7, 8, 9, 10, 14, 25)         # the stand-in vocabulary of the reference-side tokenizer
    for kw, exp in zip(C["initial"], want["initial"]):
        kw = dict(kw)
        sample_len = kw.pop("sample_len", None) or PT.WHISPER_HOST_N_CTX // 2
        got = D.initial_tokens(t, DecodingOptions(language="en", **kw), PT.WHISPER_HOST_N_CTX, sample_len)
        assert got == exp, (kw, got, exp)
    for sup, exp in zip(C["suppress"], want["suppress"]):
        assert list(D.get_suppress_tokens(t, sup)) == exp, sup
    for r, exp in zip(C["rank"], want["rank"]):
        assert D.rank_group(r["tokens"], r["sum_logprobs"], r["length_penalty"]) == exp, r
    for text, exp in zip(C["text"], want["ratio"]):
        assert abs(D.compression_ratio(text) - exp) < 1e-12
