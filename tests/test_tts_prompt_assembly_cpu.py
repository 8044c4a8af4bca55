"""Host-side prompt assembly of the autoregressive TTS families against the reference's own code: ``tests/golden/ref_qwen3_inputs.npz`` holds the outputs of
the reference's ``Model._prepare_generation_inputs`` / ``_prepare_batch_inputs`` (qwen3_tts.py:326-604, executed over the numpy stand-in for MLX by
tests/golden/make_reference_fixtures.py) with the talker's embedding tables replaced by seeded lookup tables.  This package's ``Model`` methods, given the
same tables and tokenizer, must return the same tensors exactly (the assembly only gathers, adds and concatenates)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import pt_layouts as PT  # noqa: E402


def _qwen3_host(fx):
    from mlx_audio_amd.tts.models.qwen3_tts.qwen3_tts import Model

    text_table, codec_table = torch.from_numpy(fx["text_table"]), torch.from_numpy(fx["codec_table"])
    m = Model.__new__(Model)
    m.config = PT.qwen3_input_config()
    m.tokenizer = PT.QwenCharTokenizer()
    m.talker = SimpleNamespace(device="cpu", codec_table=codec_table, embed_text=lambda ids: text_table[ids.long()])
    return m


def test_qwen3_prompt_assembly_matches_the_reference():
    fx = np.load(os.path.join(GOLD, "ref_qwen3_inputs.npz"))
    m = _qwen3_host(fx)
    for i, c in enumerate(PT.QWEN3_INPUT_CASES):
        e, tr, pad = m._prepare_generation_inputs(c["text"], language=c["language"], speaker=c["speaker"], instruct=c["instruct"])
        for name, got in (("embeds", e), ("trailing", tr), ("pad", pad)):
            want = fx[f"{name}{i}"]
            assert tuple(got.shape) == want.shape, (i, name, tuple(got.shape), want.shape)
            assert np.array_equal(got.numpy(), want), (i, name, float(np.abs(got.numpy() - want).max()))


def test_qwen3_batch_prompt_assembly_matches_the_reference():
    fx = np.load(os.path.join(GOLD, "ref_qwen3_inputs.npz"))
    m = _qwen3_host(fx)
    b = PT.QWEN3_BATCH_CASE
    bi = m._prepare_batch_inputs(b["texts"], language=b["language"], speakers=b["speakers"], instructs=b["instructs"], return_metadata=True)
    assert list(bi.left_padding) == fx["left_padding"].tolist() and list(bi.prefill_lens) == fx["prefill_lens"].tolist()
    assert list(bi.trailing_lens) == fx["trailing_lens"].tolist()
    assert np.array_equal(bi.input_embeds.numpy(), fx["batch_embeds"]) and np.array_equal(bi.trailing_text_hidden.numpy(), fx["batch_trailing"])
    assert np.array_equal(bi.tts_pad_embed.numpy(), fx["batch_pad"]) and np.array_equal(bi.attention_mask.numpy(), fx["batch_mask"])
    x, tr, pad, mask = m._prepare_batch_inputs(b["texts"], language=b["language"], speakers=b["speakers"], instructs=b["instructs"])
    assert np.array_equal(x.numpy(), fx["batch_embeds"]) and np.array_equal(mask.numpy(), fx["batch_mask"])


def _qwen3_clone_host(fx, xvec=True, kind="base", enc=True):
    """This package's ``Model`` over the scripted parts the reference ran on (pt_layouts): seeded tables, the stand-in codec and x-vector."""
    from mlx_audio_amd.tts.models.qwen3_tts.qwen3_tts import Model

    text_table, codec_table, cp_tables = (torch.from_numpy(fx[k]) for k in ("text_table", "codec_table", "cp_tables"))
    H = text_table.shape[1]

    def embed_codes(codes):   # [1, T, groups] -> [1, T, H]: group 0 from the talker's table, group i from the code predictor's table i - 1, summed in order
        c = codes.long()
        out = codec_table[c[..., 0]]
        for i in range(cp_tables.shape[0]):
            out = out + cp_tables[i][c[..., i + 1]]
        return out

    class Tok:
        has_encoder = enc is not False
        decoder = SimpleNamespace(device="cpu")

        def encode(self, audio):
            return torch.from_numpy(PT.qwen3_fake_codes(np.asarray(audio)))

        def decode(self, codes):
            a, n = PT.qwen3_fake_decode(codes.numpy())
            return torch.from_numpy(a), torch.from_numpy(n)

    m = Model.__new__(Model)
    m.config = PT.qwen3_icl_config(kind)
    m.tokenizer = PT.QwenCharTokenizer()
    m.talker = SimpleNamespace(device="cpu", codec_table=codec_table, embed_text=lambda ids: text_table[ids.long()], embed_codes=embed_codes)
    m.speech_tokenizer = None if enc is None else Tok()
    m.speaker_encoder = object() if xvec else None
    m.extract_speaker_embedding = lambda audio, sr=24000: torch.from_numpy(PT.qwen3_fake_xvector(np.asarray(audio), H))
    m._icl_cache = {}
    m._sample_rate = 24000
    return m


def _clip(spec):
    n, s, lead = spec
    return torch.from_numpy(PT.qwen3_fake_clip(n, s).reshape((1,) * lead + (n,)))


def test_qwen3_in_context_prompts_match_the_reference():
    """``ref_qwen3_icl.npz`` = the reference's ``_prepare_icl_generation_inputs`` (qwen3_tts.py:606-803), the in-context branch of ``_prepare_batch_inputs``
    (:509-527) and ``_prepare_generation_inputs`` with a clip but no transcript (:383-384) on scripted parts: same tensors, exactly; same clip cache."""
    fx = np.load(os.path.join(GOLD, "ref_qwen3_icl.npz"))
    hosts = {True: _qwen3_clone_host(fx, True), False: _qwen3_clone_host(fx, False)}
    for i, c in enumerate(PT.QWEN3_ICL_CASES):
        e, tr, pad, codes = hosts[c["xvec"]]._prepare_icl_generation_inputs(c["text"], ref_audio=_clip(c["clip"]), ref_text=c["ref_text"], language=c["language"])
        for name, got in (("icl_embeds", e), ("icl_trailing", tr), ("icl_pad", pad)):
            want = fx[f"{name}{i}"]
            assert tuple(got.shape) == want.shape, (i, name, tuple(got.shape), want.shape)
            assert np.array_equal(got.numpy(), want), (i, name, float(np.abs(got.numpy() - want).max()))
        assert np.array_equal(np.asarray(codes), fx[f"icl_codes{i}"])
    assert [len(hosts[True]._icl_cache), len(hosts[False]._icl_cache)] == fx["cache_entries"].tolist()   # cases 0 and 1 share one entry
    b = PT.QWEN3_ICL_BATCH
    bi = _qwen3_clone_host(fx)._prepare_batch_inputs(b["texts"], language=b["language"], ref_audio=_clip(b["clip"]), ref_text=b["ref_text"], return_metadata=True)
    assert list(bi.left_padding) == fx["left_padding"].tolist() and list(bi.prefill_lens) == fx["prefill_lens"].tolist()
    assert list(bi.trailing_lens) == fx["trailing_lens"].tolist() == [1, 1, 1]        # in-context prompts carry all their text in the prefill
    assert np.array_equal(bi.input_embeds.numpy(), fx["batch_embeds"]) and np.array_equal(bi.trailing_text_hidden.numpy(), fx["batch_trailing"])
    assert np.array_equal(bi.attention_mask.numpy(), fx["batch_mask"]) and np.array_equal(np.asarray(bi.ref_codes), fx["batch_ref_codes"])
    for i, c in enumerate(PT.QWEN3_XVEC_CASES):
        e, tr, _ = _qwen3_clone_host(fx, c["xvec"])._prepare_generation_inputs(c["text"], language=c["language"], speaker=c["speaker"], ref_audio=_clip(c["clip"]))
        assert np.array_equal(e.numpy(), fx[f"xvec_embeds{i}"]) and np.array_equal(tr.numpy(), fx[f"xvec_trailing{i}"]), i
    # the x-vector really sits in the speaker slot: with and without a speaker encoder the prompts differ in exactly that position (case 1 vs 2)
    d = np.abs(fx["xvec_embeds1"] - fx["xvec_embeds2"]).max(axis=(0, 2))
    assert (d > 0).sum() == 1


def test_qwen3_decode_behind_reference_codes_and_shared_reference_rules_match_the_reference(monkeypatch):
    """``_decode_icl_generated_codes`` (qwen3_tts.py:1085-1112: reference frames in front, trim to the valid length, cut ``ref_len / total_len`` of the
    samples), ``_normalize_shared_batch_refs`` (:1582-1649: outcomes and error texts) and ``supports_tts_batch`` with references (:233-244)."""
    import json

    fx = np.load(os.path.join(GOLD, "ref_qwen3_icl.npz"))
    m = _qwen3_clone_host(fx)
    for i, (n_gen, n_ref) in enumerate(PT.QWEN3_ICL_DECODE_CASES):
        gen, ref = PT.qwen3_icl_decode_case(n_gen, n_ref)
        audio = m._decode_icl_generated_codes(torch.from_numpy(gen), torch.from_numpy(ref))
        want = fx[f"decoded{i}"]
        assert tuple(audio.shape) == want.shape and np.array_equal(audio.numpy(), want), (i, tuple(audio.shape), want.shape)
    tables = json.loads(str(fx["tables"]))
    import mlx_audio_amd.utils as U

    monkeypatch.setattr(U, "load_audio", lambda path, sample_rate=None: "loaded:" + str(path))
    got = [PT.qwen3_shared_ref_outcome(m._normalize_shared_batch_refs, c) for c in PT.QWEN3_SHARED_REF_CASES]
    assert got == tables["shared"]
    sup = [bool(_qwen3_clone_host(fx, True, c["kind"], c["enc"]).supports_tts_batch(**c["kw"])) for c in PT.QWEN3_SUPPORTS_BATCH_CASES]
    assert sup == tables["supports"] and any(sup) and not all(sup)


def test_csm_prompt_frames_and_generate_bookkeeping_match_the_reference():
    """``ref_csm_generate.json`` = the reference's CSM ``Model.generate`` (sesame.py:730-866) with its ``_tokenize_*`` builders on scripted parts
    (make_reference_fixtures.run_csm_generate).  With the same stand-ins this package's ``generate`` hands the engine the same prompt frames and masks
    (speaker prefix, prompt splitting, context with / without voice matching, reference audio as the first segment, EOS frame rule), honours the same
    frame budget and yields results holding the same frames (one per prompt; per streaming interval with ``stream=True``)."""
    import json

    from mlx_audio_amd.tts.models.sesame.sesame import Model, Segment

    want = json.load(open(os.path.join(GOLD, "ref_csm_generate.json")))
    K = PT.CSM_CODEBOOKS
    tok = PT.CsmCharTokenizer()
    assert len(want) == len(PT.CSM_GENERATE_CASES)
    for case, exp in zip(PT.CSM_GENERATE_CASES, want):
        prompts = []

        class Engine:
            device = "cpu"

            def generate(self, tokens, mask, max_frames, **kw):
                i = len(prompts)
                prompts.append(dict(tokens=tokens[0].to(torch.int64).tolist(), mask=mask[0].to(torch.int64).tolist(), max_frames=max_frames))
                n = min(case["frames"][i], max_frames)
                return {"frames": [torch.tensor([PT.csm_frame(i, j) for j in range(n)], dtype=torch.int32).reshape(n, K)]}

            def generate_chunks(self, tokens, mask, max_frames, *, chunk, **kw):   # the streaming frame loop: the same frames, a block per interval
                fr = self.generate(tokens, mask, max_frames, **kw)["frames"][0].to(torch.int64)
                for j in range(0, fr.shape[0], chunk):
                    yield fr[None, j:j + chunk]

        class Host(Model):
            def _decode_frames(self, frames):
                return torch.zeros(frames.shape[0] * 1920)

            def generate_result(self, samples, start_time, stream=False, audio=None):
                return dict(n=int(samples.shape[0]), stream=bool(stream), frames=samples.to(torch.int64).tolist())

        m = Host.__new__(Host)
        m._frame_size, m._sample_rate, m.model = K + 1, 24000, Engine()
        m._speaker_prefix_space, m._default_voice_match = case["cfg"]["speaker_prefix_space"], case["cfg"]["voice_match"]
        m._use_default_voice_prompt = False
        m._text_tokenizer = SimpleNamespace(encode=tok.ids)
        m._audio_tokenizer = SimpleNamespace(encode=lambda x: torch.from_numpy(PT.csm_fake_codes(np.asarray(x)[0, 0]))[None],
                                             new_stream=lambda b: SimpleNamespace(batch=b),      # what MimiStreamingDecoder drives (stream=True)
                                             decode_step=lambda tokens, st: torch.zeros(tokens.shape[0], 1, tokens.shape[2] * 1920))
        kw = dict(case["kw"])
        if "context" in kw:
            kw["context"] = [Segment(speaker=sp, text=t, audio=torch.from_numpy(PT.csm_audio(*au))) for sp, t, au in kw["context"]]
        if "ref_audio" in kw:
            kw["ref_audio"] = torch.from_numpy(PT.csm_audio(*kw["ref_audio"]))
        results = list(m.generate(case["text"], **kw))
        assert len(prompts) == len(exp["prompts"]), case["name"]
        for got, ref in zip(prompts, exp["prompts"]):
            first = ref["calls"][0]                      # the reference's first generate_frame call of a prompt carries the whole prompt
            assert got["tokens"] == first["tokens"] and got["mask"] == first["mask"], case["name"]
            assert first["pos"] == list(range(len(first["tokens"])))
            # later calls: one frame [sample, 0] with the text slot masked out, at the next position (the engine's step convention)
            for j, call in enumerate(ref["calls"][1:]):
                assert call["mask"] == [[1] * K + [0]] and call["pos"] == [len(first["tokens"]) + j] and call["tokens"][0][-1] == 0
            assert len(ref["calls"]) <= got["max_frames"]
        assert results == exp["results"], (case["name"], results, exp["results"])


def test_csm_audio_context_without_an_encoder_is_refused():
    from mlx_audio_amd.tts.models.sesame.sesame import Model, Segment

    m = Model.__new__(Model)
    m._frame_size, m._audio_tokenizer, m.model = 5, None, object()
    m._use_default_voice_prompt, m._default_voice_match = False, True
    import pytest

    with pytest.raises(NotImplementedError, match="Mimi encoder"):
        list(m.generate("hi", context=[Segment(0, "x", torch.zeros(10))]))
    with pytest.raises(NotImplementedError, match="Mimi encoder"):
        list(m.generate("hi", ref_audio=torch.zeros(10), ref_text="x"))
