"""The record types that cross the drop-in boundary have the reference's fields, in the reference's order, with the reference's defaults
(``tests/golden/ref_dataclasses.json`` = ``dataclasses.fields`` of the reference's own classes, written by tests/golden/make_reference_fixtures.py)."""
import dataclasses
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fields(cls):
    out = []
    for f in dataclasses.fields(cls):
        d = f.default if f.default is not dataclasses.MISSING else ("<factory>" if f.default_factory is not dataclasses.MISSING else "<required>")
        out.append([f.name, d if isinstance(d, (int, float, str, bool, type(None))) else repr(d)])
    return out


def test_boundary_records_have_the_reference_fields():
    from mlx_audio_amd import server_inference as srv
    from mlx_audio_amd.stt.models import base as stt
    from mlx_audio_amd.stt.models.whisper import decoding as dec
    from mlx_audio_amd.tts import continuous as cont
    from mlx_audio_amd.tts.models import base

    want = json.load(open(os.path.join(GOLD, "ref_dataclasses.json")))
    got = {"GenerationResult": base.GenerationResult, "BatchGenerationResult": base.BatchGenerationResult, "TTSBatchOptions": cont.TTSBatchOptions,
           "TTSBatchItem": cont.TTSBatchItem, "TTSBatchEvent": cont.TTSBatchEvent, "InferenceResultChunk": srv.InferenceResultChunk,
           "InferenceContext": srv.InferenceContext, "InferenceRequest": srv.InferenceRequest, "DecodingOptions": dec.DecodingOptions,
           "DecodingResult": dec.DecodingResult, "STTOutput": stt.STTOutput}
    assert set(got) == set(want)
    for name, cls in got.items():
        mine, ref = _fields(cls), want[name]
        ref_names = [n for n, _ in ref]
        mine_d = dict((n, d) for n, d in mine)
        missing = [n for n in ref_names if n not in mine_d]
        assert not missing, (name, "missing fields", missing)
        # same order for the reference's fields (positional construction), same defaults; extra fields may only follow them
        assert [n for n, _ in mine][:len(ref_names)] == ref_names, (name, [n for n, _ in mine], ref_names)
        for n, d in ref:
            both_nan = isinstance(d, float) and d != d and isinstance(mine_d[n], float) and mine_d[n] != mine_d[n]
            assert both_nan or mine_d[n] == d, (name, n, mine_d[n], d)
