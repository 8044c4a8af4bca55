"""Host schedules of the Qwen3-TTS cloning path with the device pieces replaced by their DEFINITIONS (torch on the CPU), so that the wiring is checked
where there is no GPU:
  * ``Qwen3TTSSpeakerEncoder.__call__`` -- which channel slices feed which conv, where the reflect padding / "chunk + previous output" / gate +
    residual passes go, how the multi-layer concatenation and the [x | mean | std] pooling input are laid out -- against the reference's own module run
    (tests/golden/ref_qwen3_speaker_encoder.npz);
  * ``Model.generate`` / ``batch_generate`` with ``ref_audio`` + ``ref_text`` over a scripted frame loop and a scripted codec: routing, the repetition-penalty
    floor, per-sequence frame budgets, decode behind the reference codes, streaming chunks.
The kernels themselves are tested on the device (tests/test_qwen3_clone_gpu.py); nothing here is a product path."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import pt_layouts as PT  # noqa: E402


def _reflect_index(T, pad):
    idx = torch.arange(-pad, T + pad).abs()
    return torch.where(idx >= T, 2 * (T - 1) - idx, idx)


class _DefOps:
    """The definitions of the launches the speaker encoder issues (see include/mi355audio.h), on whatever tensors / views they are handed."""
    ACT_LEAKY = 1
    calls = []

    @staticmethod
    def require_gpu():
        return None

    @staticmethod
    def pack_conv(w, b, dev, f16=False):
        return SimpleNamespace(w=w, bias=b, cout=w.shape[0], k=w.shape[1], cin=w.shape[2])

    @classmethod
    def conv_gemm(cls, x, pc, y, *, dil=1, pad=0, post_act=0, post_slope=0.0, precision=2):
        assert x.shape[2] == pc.cin and y.shape[2] == pc.cout and pad == 0
        out = F.conv1d(x.transpose(1, 2), pc.w.permute(0, 2, 1), pc.bias, dilation=dil).transpose(1, 2)
        assert out.shape == y.shape, (tuple(out.shape), tuple(y.shape))      # the padded input has exactly the rows the taps need
        if post_act == cls.ACT_LEAKY:
            out = F.leaky_relu(out, post_slope)
        else:
            assert post_act == 0
        cls.calls.append(("conv", pc.cin, pc.cout, pc.k, dil))
        y.copy_(out)
        return y

    @classmethod
    def ecapa_rows(cls, x, y, *, pad=0, gate=None, res=None, pre_tanh=False):
        B, T, C = x.shape
        assert tuple(y.shape) == (B, T + 2 * pad, C) and pad < T
        idx = _reflect_index(T, pad)
        v = torch.tanh(x) if pre_tanh else x
        if gate is not None:
            assert tuple(gate.shape) == (B, C)
            v = v * torch.sigmoid(gate)[:, None, :]
        if res is not None:
            assert tuple(res.shape) == (B, T, C)
            v = v + res
        cls.calls.append(("rows", C, pad, gate is not None, res is not None, pre_tanh))
        y.copy_(v[:, idx])
        return y

    @staticmethod
    def time_moments(x, eps=0.0, want_std=True):
        return x.mean(1), (torch.sqrt(x.var(1, unbiased=False) + eps) if want_std else None)

    @staticmethod
    def attentive_pool(x, logits, eps=1e-12):
        w = torch.softmax(logits, 1)
        m = (w * x).sum(1)
        return torch.cat([m, torch.sqrt((w * (x - m[:, None]) ** 2).sum(1).clamp_min(eps))], 1)

    @staticmethod
    def broadcast_rows(v, y, lens=None):
        y.copy_(v[:, None, :].expand_as(y))
        return y


def test_speaker_encoder_schedule_reproduces_the_reference_run(monkeypatch):
    from mlx_audio_amd.tts.models.qwen3_tts import speaker_encoder as SE

    _DefOps.calls = []
    monkeypatch.setattr(SE, "ops", _DefOps)
    monkeypatch.setattr(SE, "make_lin", lambda w, b, dev: SimpleNamespace(w=w, b=b, rm=SimpleNamespace(n=w.shape[0], k=w.shape[1])))

    def linear(x, l, y, *, post_act=0, precision=2):
        out = x @ l.w.t() + l.b
        y.copy_(F.leaky_relu(out, 0.0) if post_act == _DefOps.ACT_LEAKY else out)
        return y

    monkeypatch.setattr(SE, "linear", linear)
    fx = np.load(os.path.join(GOLD, "ref_qwen3_speaker_encoder.npz"))
    c = SE.tiny_speaker_encoder_config()
    w = SE.make_speaker_encoder_weights(c, seed=int(fx["seed_w"]))
    mels = SE.make_mels(2, int(fx["frames"]), c.mel_dim, seed=int(fx["seed_mel"]))
    enc = SE.Qwen3TTSSpeakerEncoder(c, w, device="cpu")
    st = {}
    emb = enc(mels, stages=st)
    want = fx["embedding"]
    err, peak = float(np.abs(emb.numpy() - want).max()), float(np.abs(want).max())
    print(f"speaker encoder schedule (definitions on the CPU) vs the reference run: {err:.2e} (peak {peak:.2f})")
    assert emb.shape == want.shape and err < 2e-5 * peak
    # stage by stage against the restatement too (a wrong slice that happened to cancel would show here)
    from oracle.ecapa_ref import EcapaRef

    ost = {}
    EcapaRef(w, c)(mels, ost)
    for k in ("block1", "block2", "block3", "mfa", "asp_logits"):
        assert float((st[k] - ost[k]).abs().max()) < 2e-5 * float(ost[k].abs().max()), k
    assert float((st["pooled"] - ost["pooled"][:, 0]).abs().max()) < 2e-5 * float(ost["pooled"].abs().max())
    # the launch list: per SE-Res2Net block 2 k = 1 convs + (scale - 1) dilated k = 3 convs on chunk-wide slices, one gate + residual pass; no pass adds a
    # residual to chunk 1, every later chunk does
    sub = c.enc_channels[1] // c.enc_res2net_scale
    convs = [x for x in _DefOps.calls if x[0] == "conv"]
    assert convs.count(("conv", sub, sub, 3, 2)) == convs.count(("conv", sub, sub, 3, 3)) == convs.count(("conv", sub, sub, 3, 4)) == c.enc_res2net_scale - 1
    rows = [x for x in _DefOps.calls if x[0] == "rows"]
    assert sum(1 for r in rows if r[3]) == 3 and all(r[4] for r in rows if r[3])              # three gate passes, each with the block residual
    assert sum(1 for r in rows if r[4] and not r[3]) == 3 * (c.enc_res2net_scale - 2)         # chunk + previous output: every chunk from the third on
    assert sum(1 for r in rows if r[2] > 0 and not r[3] and not r[4]) == 3 + 1                  # plain reflect padding: chunk 1 of each block + the first TDNN
    assert sum(1 for r in rows if r[5]) == 1                                                   # one tanh pass (the attention)
    # argument checks of the engine
    with pytest.raises(ValueError):
        enc(mels[:, :, :5])
    with pytest.raises(ValueError):
        enc(mels[:, :4])
    with pytest.raises(ValueError):
        SE.Qwen3TTSSpeakerEncoder(SE.Qwen3TTSSpeakerEncoderConfig(enc_channels=[64, 64, 96, 128]), device="cpu")
    with pytest.raises(ValueError):
        SE.Qwen3TTSSpeakerEncoder(c, {**w, "blocks.9.conv.weight": torch.zeros(1, 1, 1)}, device="cpu")


# ------------------------------------------------------------------------------------------------ Model.generate / batch_generate with a reference clip
def _scripted_model(monkeypatch, frames_of, xvec=True):
    """``Model`` over scripted parts: embedding tables, a frame loop that returns ``frames_of(batch index, prefill length)`` frames of recognisable codes
    and records how it was called, the stand-in codec of pt_layouts."""
    from mlx_audio_amd.tts.models.qwen3_tts.qwen3_tts import Model

    g = torch.Generator().manual_seed(0)
    H, G = 8, PT.QWEN3_ICL_GROUPS
    text_table = torch.randn(PT.QWEN3_TEXT_VOCAB, H, generator=g)
    codec_table = torch.randn(PT.QWEN3_CODEC_VOCAB, H, generator=g)
    calls = []

    class Engine:
        device = "cpu"
        talker = SimpleNamespace(cos=torch.zeros(4096, 4))

        def __init__(self):
            self.codec_table = codec_table

        def embed_text(self, ids):
            return text_table[ids.long()]

        def embed_codes(self, codes):
            return codec_table[codes.long() % PT.QWEN3_CODEC_VOCAB].sum(2)

        def generate(self, prefill, trailing, tts_pad, max_frames, *, temperature, top_k, top_p, repetition_penalty, left_pad=None, generator=None, **kw):
            B = prefill.shape[0]
            n = [min(frames_of(b, prefill.shape[1]), max_frames) for b in range(B)]
            T = max(n)
            codes = torch.zeros((B, T, G), dtype=torch.int64)
            for b in range(B):
                codes[b, : n[b]] = 1 + (torch.arange(n[b])[:, None] * 3 + torch.arange(G)[None] + 5 * b) % 30
            calls.append(dict(prefill=tuple(prefill.shape), trailing=tuple(trailing.shape), max_frames=max_frames, repetition_penalty=repetition_penalty,
                              left_pad=None if left_pad is None else left_pad.tolist(), kw=sorted(kw)))
            return dict(codes=codes, finished_at=torch.tensor([x if x < max_frames else -1 for x in n]))

        def generate_iter(self, prefill, trailing, tts_pad, max_frames, *, chunk=0, **kw):
            """the engine's streaming contract: blocks of ``chunk`` frames (EOS frame excluded) while the loop runs, then the final dict"""
            out = self.generate(prefill, trailing, tts_pad, max_frames, **kw)
            calls[-1]["chunk"] = chunk
            fa = int(out["finished_at"][0])
            n = fa if fa >= 0 else out["codes"].shape[1]
            if chunk:
                for s0 in range(0, n, chunk):
                    e = min(s0 + chunk, n)
                    streamed.append((s0, e))
                    yield dict(block=out["codes"][:, s0:e], first_frame=s0, last=e == n)
            yield out

    streamed = []

    class Tok:
        has_encoder = True
        decode_upsample_rate = PT.QWEN3_ICL_UP
        decoder = SimpleNamespace(device="cpu")

        def encode(self, audio):
            return torch.from_numpy(PT.qwen3_fake_codes(np.asarray(audio)))

        def decode(self, codes):
            a, n = PT.qwen3_fake_decode(codes.numpy())
            return torch.from_numpy(a), torch.from_numpy(n)

    def decoder(codes):           # [1, groups, T] -> [1, 1, T * up]: the streaming path decodes chunks with left context through the decoder itself
        a, _ = PT.qwen3_fake_decode(codes.permute(0, 2, 1).numpy())
        return torch.from_numpy(a)[:, None, :]

    class Dec:   # the codec decoder's surface: one-shot call + the carried-state streaming pair (the stand-in codec has no memory, so the state only counts)
        device = "cpu"
        __call__ = staticmethod(decoder)

        def new_stream(self, batch=1):
            return SimpleNamespace(frames=0, batch=batch)

        def streaming_step(self, codes, st):
            st.frames += codes.shape[2]
            return decoder(codes)

    tok = Tok()
    tok.decoder = Dec()
    m = Model.__new__(Model)
    m.config = PT.qwen3_icl_config("base")
    m._sample_rate = 24000
    m.tokenizer = PT.QwenCharTokenizer()
    m.talker = Engine()
    m.speech_tokenizer = tok
    m.speaker_encoder = object() if xvec else None
    m.extract_speaker_embedding = lambda audio, sr=24000: torch.from_numpy(PT.qwen3_fake_xvector(np.asarray(audio), H))
    m._icl_cache = {}
    m.supported_speakers = ["vivian"]
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda *a, **k: 0)
    return m, calls


def _expected_audio(ref_codes, gen):
    """qwen3_tts.py:1085-1112 by hand on the stand-in codec."""
    full = np.concatenate([np.transpose(ref_codes, (0, 2, 1)), gen[None]], axis=1)
    audio, lengths = PT.qwen3_fake_decode(full)
    audio = audio[0]
    valid = int(lengths[0])
    if 0 < valid < len(audio):
        audio = audio[:valid]
    cut = int(ref_codes.shape[2] / full.shape[1] * len(audio))
    return audio[cut:] if 0 < cut < len(audio) else audio


def test_generate_routes_a_clip_with_transcript_through_the_in_context_path(monkeypatch):
    m, calls = _scripted_model(monkeypatch, lambda b, L: 5)
    clip = torch.from_numpy(PT.qwen3_fake_clip(500, 4))
    ref_codes = PT.qwen3_fake_codes(clip.numpy())
    res = list(m.generate("Say this.\nAnd this on a second line", ref_audio=clip, ref_text="the clip's words", temperature=0.0, max_tokens=40, repetition_penalty=1.05))
    assert len(res) == 1 and len(calls) == 1                      # ONE segment: the in-context path does not split the text (qwen3_tts.py:2226-2235)
    assert calls[0]["repetition_penalty"] == 1.5 and calls[0]["max_frames"] == 40 and calls[0]["trailing"] == (1, 1, 8) and calls[0]["left_pad"] is None
    gen = (1 + (np.arange(5)[:, None] * 3 + np.arange(PT.QWEN3_ICL_GROUPS)[None]) % 30).astype(np.int64)
    want = _expected_audio(ref_codes, gen)
    assert res[0].token_count == 5 and res[0].samples == len(want) and np.array_equal(res[0].audio.numpy(), want)
    assert res[0].segment_idx == 0 and not res[0].is_streaming_chunk
    # a penalty above the floor is kept
    list(m.generate("x", ref_audio=clip, ref_text="t", repetition_penalty=2.0, max_tokens=3))
    assert calls[-1]["repetition_penalty"] == 2.0
    # streaming: chunks of the generated frames only (2 frames per 0.16 s), the last one flagged
    chunks = list(m.generate("Say this.", ref_audio=clip, ref_text="the clip's words", max_tokens=40, stream=True, streaming_interval=0.16))
    assert [r.token_count for r in chunks] == [2, 2, 1] and [r.is_final_chunk for r in chunks] == [False, False, True]
    assert all(r.is_streaming_chunk and r.samples == r.token_count * PT.QWEN3_ICL_UP for r in chunks)
    assert calls[-1]["chunk"] == 2     # the frame loop itself ran in streaming mode: blocks left it while it ran (qwen3_tts.py:1426-1465)
    # no frames at all (EOS on the first one): nothing is yielded
    m0, _ = _scripted_model(monkeypatch, lambda b, L: 0)
    assert list(m0.generate("x", ref_audio=clip, ref_text="t", max_tokens=4)) == []
    # without a transcript the clip only contributes its x-vector: the plain per-line loop, one prompt position longer
    m2, calls2 = _scripted_model(monkeypatch, lambda b, L: 2)
    r2 = list(m2.generate("one\ntwo", ref_audio=clip, max_tokens=9))
    assert len(r2) == 2 and [r.segment_idx for r in r2] == [0, 1] and all(c["repetition_penalty"] == 1.05 for c in calls2)
    m3, calls3 = _scripted_model(monkeypatch, lambda b, L: 2)
    list(m3.generate("one", max_tokens=9))
    assert calls2[0]["prefill"][1] == calls3[0]["prefill"][1] + 1
    # a tokenizer without its encoder half: transcript or not, the clip is x-vector only (qwen3_tts.py:1227-1231)
    m2.speech_tokenizer.has_encoder = False
    n_before = len(calls2)
    assert len(list(m2.generate("one", ref_audio=clip, ref_text="t", max_tokens=9))) == 1 and calls2[-1]["repetition_penalty"] == 1.05 and len(calls2) == n_before + 1
    # other model types never look at the clip
    m4, calls4 = _scripted_model(monkeypatch, lambda b, L: 2)
    m4.config.tts_model_type = "custom_voice"
    list(m4.generate("one", voice="vivian", ref_audio=clip, ref_text="t", max_tokens=9))
    m4.config.tts_model_type = "base"
    list(m4.generate("one", voice="vivian", max_tokens=9))
    assert calls4[0]["prefill"] == calls4[1]["prefill"] and calls4[0]["repetition_penalty"] == 1.05
    with pytest.raises(ValueError):
        list(m4.generate("one", voice="nobody"))


def test_batch_generate_with_a_shared_reference(monkeypatch):
    texts = ["Short.", "A considerably longer second sentence for the batch.", "Mid length text"]
    n_frames = {0: 100, 1: 7, 2: 200}
    m, calls = _scripted_model(monkeypatch, lambda b, L: n_frames[b])
    clip = torch.from_numpy(PT.qwen3_fake_clip(500, 4))
    ref_codes = PT.qwen3_fake_codes(clip.numpy())
    out = list(m.batch_generate(texts, ref_audios=[clip, clip, clip], ref_texts=["words"] * 3, max_tokens=150, repetition_penalty=1.1))
    caps = [min(150, max(75, len(m.tokenizer.encode(t)) * 6)) for t in texts]
    assert caps[0] == 75 and caps[1] == 150                          # 6 frames per text token, at least 75 (qwen3_tts.py:1823-1827)
    assert len(calls) == 1 and calls[0]["repetition_penalty"] == 1.5 and calls[0]["max_frames"] == max(caps)
    assert calls[0]["left_pad"] is not None and min(calls[0]["left_pad"]) == 0 and calls[0]["trailing"] == (3, 1, 8)
    assert [r.sequence_idx for r in out] == [0, 1, 2]
    want_n = [min(n_frames[b], caps[b]) for b in range(3)]
    assert [r.token_count for r in out] == want_n and want_n[0] == 75 and want_n[1] == 7      # row 0 stops at its own budget, row 2 at its cap of the shared loop
    for b, r in enumerate(out):
        gen = (1 + (np.arange(want_n[b])[:, None] * 3 + np.arange(PT.QWEN3_ICL_GROUPS)[None] + 5 * b) % 30).astype(np.int64)
        want = _expected_audio(ref_codes, gen)
        assert r.samples == len(want) and np.array_equal(r.audio.numpy(), want), b
    # the refusals of the reference (qwen3_tts.py:1726-1740) and of the shared-reference rule
    with pytest.raises(ValueError, match="does not support voices"):
        list(m.batch_generate(texts, ref_audio=clip, ref_text="w", voices=["vivian", None, None]))
    with pytest.raises(ValueError, match="does not support instructs"):
        list(m.batch_generate(texts, ref_audio=clip, ref_text="w", instructs=[None, "slow", None]))
    with pytest.raises(ValueError, match="requires both"):
        list(m.batch_generate(texts, ref_audio=clip))
    with pytest.raises(ValueError, match="only one shared"):
        list(m.batch_generate(texts, ref_audios=[clip, clip.clone(), clip], ref_texts=["w"] * 3))
    m.speech_tokenizer.has_encoder = False
    with pytest.raises(ValueError, match="requires a speech tokenizer encoder"):
        list(m.batch_generate(texts, ref_audio=clip, ref_text="w"))


# ------------------------------------------------------------------------------------------------ batch_generate(stream=True) on a scripted slot engine
class _ScriptedSlots:
    """The interface of ``Qwen3TalkerSlots`` with every row's life fixed in advance: row b produces ``n[b]`` frames, then EOS."""

    def __init__(self, n):
        self.n, self.f = list(n), [0] * len(n)
        self.limit = [10 ** 9] * len(n)
        self.released = []

    def admit(self, slots, input_embeds, left_pad, trailing, tts_pad):
        assert slots == list(range(len(self.n))) and input_embeds.shape[0] == len(self.n)
        out = []
        for b in slots:
            fin = self.n[b] <= 0
            self.f[b] = 0 if fin else 1
            out.append(fin)
        return out

    def advance(self, n_rows):
        out = []
        for b in range(n_rows):
            fin = self.f[b] >= self.n[b]
            if not fin:
                self.f[b] += 1
            out.append(fin)
        return out

    def take_codes(self, b, n):
        assert n <= self.f[b]
        return (1 + (torch.arange(n)[:, None] * 3 + torch.arange(PT.QWEN3_ICL_GROUPS)[None] + 5 * b) % 30).to(torch.int64)

    def release(self, b):
        self.released.append(b)


def _reference_stream_events(n, caps, max_tokens, chunk, use_icl):
    """qwen3_tts.py:1860-2010 restated as bookkeeping only: (sequence, new frames, context frames, final flag) in emission order."""
    B = len(n)
    gen, dec, fin, ev = [0] * B, [0] * B, [False] * B, []
    for _ in range(max_tokens):
        fin = [fin[b] or gen[b] >= n[b] for b in range(B)]           # the sampled token is EOS once the row has said everything
        if all(fin):
            break
        for b in range(B):
            if not fin[b]:
                gen[b] += 1
        if use_icl:
            fin = [fin[b] or gen[b] >= caps[b] for b in range(B)]
            if all(fin):
                break
        for b in range(B):
            if gen[b] and gen[b] - dec[b] >= chunk:
                ev.append((b, gen[b] - dec[b], 0 if dec[b] == 0 else min(25, dec[b]), False))
                dec[b] = gen[b]
    for b in range(B):
        if gen[b] and gen[b] > dec[b]:
            ev.append((b, gen[b] - dec[b], min(25, dec[b]), True))
            dec[b] = gen[b]
    return ev, gen


@pytest.mark.parametrize("n,max_tokens,interval,use_icl", [
    ([7, 0, 3, 64], 40, 0.16, False),      # chunk 2: a row silent from the start, one ending on a chunk boundary, one cut by max_tokens with context 25
    ([5, 9], 9, 0.4, False),               # chunk 5: both end exactly on boundaries / the budget
    ([200, 30, 200], 150, 0.3, True),      # in-context budgets (75 .. 150 frames): rows stop at their own caps; chunk 3
    ([1], 4, 2.0, False),                  # one frame, one final chunk
])
def test_batch_generate_streaming_follows_the_reference_loop(monkeypatch, n, max_tokens, interval, use_icl):
    texts = ["Short.", "A considerably longer second sentence for the batch.", "Mid length text", "four"][: len(n)]
    m, _ = _scripted_model(monkeypatch, lambda b, L: 0)
    m.speech_tokenizer.decoder = type("Dec", (), {"device": "cpu", "chunked_decode": staticmethod(
        lambda codes: torch.from_numpy(PT.qwen3_fake_decode(codes.permute(0, 2, 1).numpy())[0])[:, None, :])})()
    clip = torch.from_numpy(PT.qwen3_fake_clip(500, 4))
    ref = dict(ref_audio=clip, ref_text="words") if use_icl else {}
    caps = [min(max_tokens, max(75, len(m.tokenizer.encode(t)) * 6)) for t in texts] if use_icl else [max_tokens] * len(n)
    slots, log = _ScriptedSlots(n), {}
    out = list(m.batch_generate(texts, max_tokens=max_tokens, stream=True, streaming_interval=interval, slots=slots, codes_log=log, **ref))
    chunk = max(1, int(interval * 12.5))
    want, gen = _reference_stream_events(n, caps, max_tokens, chunk, use_icl)
    assert [(r.sequence_idx, r.token_count, r.is_final_chunk) for r in out] == [(b, new, final) for b, new, _, final in want]
    assert all(r.is_streaming_chunk for r in out) and sorted(slots.released) == list(range(len(n)))
    assert [log[b].shape[0] for b in range(len(n))] == gen
    pos = [0] * len(n)
    for r, (b, new, ctx, _) in zip(out, want):                          # the audio: the window decoded behind its context, the context's samples cut off
        codes = slots.take_codes(b, pos[b] + new)[pos[b] - ctx:]
        wav = PT.qwen3_fake_decode(codes[None].numpy())[0][0][ctx * PT.QWEN3_ICL_UP:]
        assert r.samples == new * PT.QWEN3_ICL_UP and np.array_equal(r.audio.numpy(), wav), (b, pos[b], new, ctx)
        pos[b] += new
    assert pos == gen


@pytest.mark.parametrize("prime", [False, True])
def test_icl_streaming_decoder_state_is_fresh_like_the_reference(monkeypatch, prime):
    """qwen3_tts.py:2266-2444: the reference's in-context STREAM path resets the decoder's streaming state and feeds ``streaming_step`` the newly
    generated codes only.  Priming the state with the reference clip's codes is this build's opt-in deviation (``prime_stream_with_reference``)."""
    from mlx_audio_amd.tts.models.qwen3_tts.qwen3_tts import Model

    G = PT.QWEN3_ICL_GROUPS
    fed = []

    class Dec:
        def new_stream(self, n):
            fed.append("new")
            return {}

        def streaming_step(self, codes, st):
            fed.append(tuple(codes.shape))
            return torch.zeros(1, 1, codes.shape[-1] * 4)

    m = object.__new__(Model)
    m.speech_tokenizer = SimpleNamespace(decoder=Dec())
    ref_codes = torch.ones(1, G, 9, dtype=torch.int64)
    monkeypatch.setattr(m, "_prepare_icl_generation_inputs", lambda *a, **k: (torch.zeros(1, 3, 8), None, None, ref_codes), raising=False)
    monkeypatch.setattr(m, "_frame_loop", lambda *a, **k: iter([{"block": torch.ones(1, 5, G, dtype=torch.int64)},
                                                                 {"block": torch.ones(1, 2, G, dtype=torch.int64), "last": True}, {"codes": None}]), raising=False)
    monkeypatch.setattr(m, "_result", lambda wav, seg, n, dt, **kw: (int(wav.shape[0]), n, kw.get("is_final_chunk")), raising=False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    out = list(m._generate_icl("some text", torch.zeros(100), "words", stream=True, prime_stream_with_reference=prime))
    assert out == [(20, 5, False), (8, 2, True)]
    assert fed == (["new", (1, G, 9)] if prime else ["new"]) + [(1, G, 5), (1, G, 2)]
