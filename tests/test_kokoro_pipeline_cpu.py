"""Kokoro's text front (chunking, voice rows, word timestamps) against the reference's own ``KokoroPipeline``: ``tests/golden/ref_kokoro_pipeline.json``
holds what the reference's pipeline (tts/models/kokoro/pipeline.py, run by tests/golden/make_reference_fixtures.py) yields with the G2P replaced by token
streams / a letter-level stand-in and the model by a recorder.  This package's pipeline, given the same stand-ins, must yield the same chunks (graphemes,
phonemes, text index), call the model with the same phonemes, voice row and speed, and stamp the same start / end times on the tokens."""
import json
import os
import sys
from types import SimpleNamespace

import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import pt_layouts as PT  # noqa: E402


def _pipe(case, calls):
    from mlx_audio_amd.tts.models.kokoro.pipeline import KokoroPipeline

    def model(ps, ref_s, speed, return_output=True):
        calls.append(dict(ps=ps, row=int(ref_s[0]), speed=float(speed)))
        return SimpleNamespace(audio=torch.zeros(len(ps) * 10), pred_dur=torch.tensor(PT.kokoro_fake_durations(ps), dtype=torch.int32))

    pipe = KokoroPipeline(case["lang"], model=model, repo_id="repo", g2p=(lambda text: ("", PT.kokoro_token_stream(*case["tokens"]))) if "tokens" in case
                          else PT.kokoro_spanish_g2p)
    pipe.voices = {"v": torch.arange(512, dtype=torch.float32)[:, None] * torch.ones(1, 4)}
    return pipe


def test_pipeline_chunks_calls_and_timestamps_match_the_reference():
    want = json.load(open(os.path.join(GOLD, "ref_kokoro_pipeline.json")))
    assert [w["name"] for w in want] == [c["name"] for c in PT.KOKORO_PIPELINE_CASES]
    for case, exp in zip(PT.KOKORO_PIPELINE_CASES, want):
        calls = []
        pipe = _pipe(case, calls)
        if "tokens" in case:
            results = list(pipe("some text", voice="v", speed=1.25))
        else:
            kw = {"split_pattern": case["split_pattern"]} if "split_pattern" in case else {}
            results = list(pipe(case["text"], voice="v", speed=0.9, **kw))
        got = [dict(graphemes=r.graphemes, phonemes=r.phonemes, text_index=r.text_index,
                    tokens=None if r.tokens is None else [[t.text, t.phonemes, t.start_ts, t.end_ts] for t in r.tokens]) for r in results]
        assert len(got) == len(exp["results"]), (case["name"], [len(g["phonemes"]) for g in got])
        for g, e in zip(got, exp["results"]):
            assert g["graphemes"] == e["graphemes"] and g["phonemes"] == e["phonemes"] and g["text_index"] == e["text_index"], case["name"]
            assert g["tokens"] == e["tokens"], case["name"]
            g2, ps, audio = results[got.index(g)]      # the backward-compatible unpacking
            assert g2 == g["graphemes"] and ps == g["phonemes"] and audio.numel() == 10 * len(ps)
        assert calls == exp["calls"], case["name"]
        assert all(len(c["ps"]) <= 510 and c["row"] == len(c["ps"]) - 1 for c in calls)
        if "from_tokens" in exp:
            calls2 = []
            pipe2 = _pipe(case, calls2)
            res2 = list(pipe2.generate_from_tokens(PT.kokoro_token_stream(*case["tokens"]), voice="v", speed=1.0))
            assert [dict(graphemes=r.graphemes, phonemes=r.phonemes, n_tokens=len(r.tokens), last_end=r.tokens[-1].end_ts) for r in res2] == exp["from_tokens"]
            assert calls2 == exp["from_tokens_calls"]


def test_pipeline_helpers_on_hand_made_tokens():
    """tokens_to_ps / tokens_to_text / waterfall_last / join_timestamps on cases small enough to read (pipeline.py:231-263, 360-399)."""
    from mlx_audio_amd.tts.models.kokoro.pipeline import KokoroPipeline as P

    T = PT.FakeMToken
    toks = [T("Hi", "hˈI", ""), T(",", ",", " "), T("you", "ju", ""), T("!", "!", ""), T("”", "”", " "), T("ok", "ˌOkˈA", "")]
    assert P.tokens_to_ps(toks) == "hˈI, ju!” ˌOkˈA" and P.tokens_to_text(toks) == "Hi, you!” ok"
    assert P.waterfall_last(toks, 20) == 5                     # after "!" and the closing quote that follows it
    assert P.waterfall_last(toks[:3], 20) == 2                 # only a comma to cut at
    assert P.waterfall_last(toks[:1], 20) == 1                 # nothing to cut at: everything
    assert P.waterfall_last(toks, 510 + len("hˈI, ju!”")) == 5 and P.waterfall_last(toks, 530) == len(toks)   # no cut leaves <= 510: everything
    P.join_timestamps(toks[:3], [5, 1, 2, 3, 1, 2, 2, 2, 4])    # <bos>=5, h ˈ I , ' ' j u <eos>... (one entry per phoneme and space)
    assert toks[0].start_ts == 4 / 80 and toks[0].end_ts == (4 + 2 * 6) / 80
    assert toks[1].start_ts == toks[0].end_ts and toks[2].start_ts is not None and toks[2].start_ts > toks[1].start_ts
    P.join_timestamps([], [1, 2, 3])
    P.join_timestamps(toks, [1, 2])                            # fewer than 3 durations: nothing happens
