"""Drop-in surface of the codec-LM TTS families (SURVEY section 8 row b): a synthetic checkpoint in the layout the reference's loaders read
(config.json + safetensors with the PyTorch / torchtune key names and conv layouts) on disk -> ``mlx_audio_amd.tts.utils.load_model`` ->
``Model.generate`` / ``batch_generate`` -> ``GenerationResult``s whose codes equal the CPU oracle's (integer path: bit-exact wherever the oracle's
top-2 margin is clear) and whose audio matches the oracle's codec decode.  Needs a real MI355X."""
import json
import re
from dataclasses import asdict

import pytest
import torch
from safetensors.torch import save_file

pytestmark = pytest.mark.gpu
DEV = "cuda"


class FakeTokenizer:
    """Deterministic stand-in for the HF tokenizers (no network): chat-template specials get fixed ids, every other character its own id."""
    SPECIAL = {"<|im_start|>": 1, "assistant": 2, "user": 5, "\n": 3, "<|im_end|>": 4}

    def __init__(self, vocab: int):
        self.vocab = vocab

    def encode(self, text, **kw):
        ids = []
        for piece in re.split("(" + "|".join(re.escape(k) for k in self.SPECIAL) + ")", text):
            if piece in self.SPECIAL:
                ids.append(self.SPECIAL[piece])
            else:
                ids.extend(10 + (ord(c) * 7) % (self.vocab - 20) for c in piece)
        return ids


import _margin  # noqa: E402  (tests/_margin.py: the knife-edge rule and its accounting)


def _gap(logits):
    top2 = torch.topk(logits, 2, dim=-1).values
    return top2[..., 0] - top2[..., 1]


# ------------------------------------------------------------------------------------------------ Qwen3-TTS
@pytest.fixture(scope="module")
def qwen3_ckpt(tmp_path_factory):
    from mlx_audio_amd.tts.models.qwen3_tts import synthetic as S
    from mlx_audio_amd.tts.models.qwen3_tts import talker as T
    from mlx_audio_amd.tts.models.qwen3_tts.config import Qwen3TTSTalkerCodePredictorConfig, Qwen3TTSTalkerConfig, Qwen3TTSTokenizerDecoderConfig

    root = tmp_path_factory.mktemp("qwen3_tts_tiny")
    cp = Qwen3TTSTalkerCodePredictorConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                           num_key_value_heads=1, head_dim=64, num_code_groups=4)
    tc = Qwen3TTSTalkerConfig(code_predictor_config=cp, vocab_size=1280, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                              num_attention_heads=2, num_key_value_heads=1, head_dim=128, num_code_groups=4, text_hidden_size=192,
                              text_vocab_size=500, codec_eos_token_id=1200, codec_think_id=1204, codec_nothink_id=1205, codec_think_bos_id=1206,
                              codec_think_eos_id=1207, codec_pad_id=1198, codec_bos_id=1199, codec_language_id={"english": 1210, "chinese": 1211},
                              spk_id={"vivian": [1220]}, max_position_embeddings=512)
    # every K == 1 conv keeps > 64 input channels: the reference's layout heuristic (qwen3_tts.py:123-157) cannot tell (out, in, 1) from
    # (out, K, 1) below that, which is why this decoder is wider than synthetic.tiny_codec_config
    dc = Qwen3TTSTokenizerDecoderConfig(latent_dim=128, codebook_dim=256, codebook_size=256, decoder_dim=384, hidden_size=128, intermediate_size=256,
                                        head_dim=64, num_attention_heads=2, num_hidden_layers=2, num_key_value_heads=2, num_quantizers=4,
                                        max_position_embeddings=512, upsample_rates=[4, 3], upsampling_ratios=[2, 2])
    up = 4 * 3 * 2 * 2
    cfg = dict(model_type="qwen3_tts", tts_model_type="base", talker_config=asdict(tc), tts_pad_token_id=497, tts_bos_token_id=498, tts_eos_token_id=499,
               sample_rate=24000)
    (root / "config.json").write_text(json.dumps(cfg))
    tw = T.make_talker_weights(tc, seed=3)
    save_file({"talker." + k: v.to(torch.bfloat16).contiguous() for k, v in tw.items()}, str(root / "model.safetensors"))
    st = root / "speech_tokenizer"
    st.mkdir()
    (st / "config.json").write_text(json.dumps(dict(decoder_config=asdict(dc), decode_upsample_rate=up, encode_downsample_rate=up)))
    cw = S.make_codec_decoder_weights(dc, seed=4)
    ck = {}
    for k, v in cw.items():
        if k.endswith("codebook.embed.weight"):  # checkpoint form: embedding_sum / cluster_usage (speech_tokenizer.py:1438-1447)
            base = k[: -len(".codebook.embed.weight")]
            ck[f"decoder.{base}._codebook.cluster_usage"] = torch.full((v.shape[0],), 2.0)
            ck[f"decoder.{base}._codebook.embedding_sum"] = 2.0 * v
        elif v.dim() == 3 and (("upsample" in k and ".0.conv.weight" in k) or re.search(r"decoder\.\d+\.block\.1\.conv\.weight", k)):
            ck["decoder." + k] = v.permute(2, 0, 1).contiguous()  # ConvTranspose1d: (C_out, K, C_in) here -> PyTorch (in, out, K)
        elif v.dim() == 3:
            ck["decoder." + k] = v.permute(0, 2, 1).contiguous()  # Conv1d: (out, K, in) -> PyTorch (out, in, K)
        else:
            ck["decoder." + k] = v
    ck["encoder.downsample.weight"] = torch.zeros(4, 4, 2)  # encoder keys are dropped by sanitize
    save_file({k: v.contiguous() for k, v in ck.items()}, str(st / "model.safetensors"))
    return dict(path=root, tc=tc, dc=dc, tw=tw, cw=cw, up=up)


def _qwen3_ref_inputs(ref, tc, tok, text, language="auto", speaker=None, instruct=None):
    """qwen3_tts.py:326-484 restated on the oracle's tensors (independent of the product's implementation)."""
    W = ref.w
    emb = lambda ids: ref.text_projection(W["model.text_embedding.weight"][torch.tensor([ids])])
    cod = lambda ids: W["model.codec_embedding.weight"][torch.tensor([ids])]
    text_embed = emb(tok.encode(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"))
    tts = emb([498, 499, 497])
    bos, eos, pad = tts[:, 0:1], tts[:, 1:2], tts[:, 2:3]
    lang = None if language == "auto" else tc.codec_language_id[language]
    pre = [tc.codec_nothink_id, tc.codec_think_bos_id, tc.codec_think_eos_id] if lang is None else [tc.codec_think_id, tc.codec_think_bos_id, lang, tc.codec_think_eos_id]
    parts = [cod(pre)] + ([cod(tc.spk_id[speaker])] if speaker else []) + [cod([tc.codec_pad_id, tc.codec_bos_id])]
    codec = torch.cat(parts, dim=1)
    combined = torch.cat([pad.expand(1, codec.shape[1] - 2, -1), bos], dim=1) + codec[:, :-1]
    x = torch.cat([text_embed[:, :3], combined, text_embed[:, 3:4] + codec[:, -1:]], dim=1)
    if instruct:
        x = torch.cat([emb(tok.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n")), x], dim=1)
    return x, torch.cat([text_embed[:, 4:-5], eos], dim=1), pad


def test_qwen3_tts_load_model_and_generate(qwen3_ckpt):
    from mlx_audio_amd.tts.utils import load_model
    from mlx_audio_amd.tts.models.base import GenerationResult
    from oracle.qwen3_codec_ref import Qwen3CodecDecoderRef
    from oracle.qwen3_talker_ref import Qwen3TalkerRef

    c = qwen3_ckpt
    model = load_model(c["path"], device=DEV)
    assert type(model).__name__ == "Model" and model.model_type == "qwen3_tts" and model.sample_rate == 24000
    assert model.speech_tokenizer is not None and not model.speech_tokenizer.has_encoder
    assert model.get_supported_speakers() == ["vivian"] and "english" in model.get_supported_languages()
    tok = FakeTokenizer(c["tc"].text_vocab_size)
    model.tokenizer = tok
    ref = Qwen3TalkerRef(c["tw"], c["tc"])
    cref = Qwen3CodecDecoderRef(c["cw"], c["dc"])
    text, frames = "hello there, general", 9
    # ---- prompt construction vs the restated reference
    x, tr, pad = model._prepare_generation_inputs(text, language="english", speaker="vivian")
    ex, etr, epad = _qwen3_ref_inputs(ref, c["tc"], tok, text, "english", "vivian")
    torch.cuda.synchronize()
    assert x.shape == ex.shape and tr.shape == etr.shape
    for g, e in ((x, ex), (tr, etr), (pad, epad)):
        assert float((g.cpu() - e).abs().max() / e.abs().max()) < 3e-4
    # ---- generate(): greedy codes == oracle codes (single-utterance trailing-text rule), audio == oracle codec decode of those codes
    res = list(model.generate(text, voice="vivian", lang_code="english", temperature=0.0, max_tokens=frames, some_cli_flag=True))
    assert len(res) == 1 and isinstance(res[0], GenerationResult) and res[0].sample_rate == 24000 and res[0].segment_idx == 0
    exp = ref.generate(ex, etr, epad, frames, temperature=0.0, record=True, pad_when_index_clamped=False)
    n = res[0].token_count
    fa = int(exp["finished_at"][0])
    assert n == (fa if fa >= 0 else exp["codes"].shape[1])
    assert res[0].samples == res[0].audio.shape[0] and res[0].audio_samples["samples"] == res[0].samples
    # re-run the engine on the same inputs to read the codes (generate() yields audio only), walk decisions until the first knife edge
    out = model._frame_loop(x, tr, pad, frames, temperature=0.0, top_k=50, top_p=1.0, repetition_penalty=1.05, pad_when_index_clamped=False)
    gc, ec = out["codes"][0].cpu(), exp["codes"][0]
    nf = min(gc.shape[0], ec.shape[0])
    margins = [float(_gap(exp["trace"][f][i][0])) for f in range(nf) for i in range(ec.shape[1])]
    checked = _margin.walk("qwen3_tts", gc[:nf].flatten().tolist(), ec[:nf].flatten().tolist(), margins, where="Model.generate")
    assert checked >= 4, checked
    wav = cref.chunked_decode(gc[:n].t()[None].long())[0, 0]
    valid = int((gc[:n, 0] > 0).sum()) * c["up"]  # speech_tokenizer.py:1112-1116: frames whose first code is 0 do not count as valid audio
    wav = wav[:valid] if 0 < valid < wav.shape[0] else wav
    got = res[0].audio.cpu()
    assert got.shape == wav.shape
    assert float((got - wav).abs().max()) <= 2e-3 * max(1.0, float(wav.abs().max()))
    # ---- stream=True (qwen3_tts.py:1426-1465): blocks of 0.16 s = 2 frames leave WHILE the frame loop runs, decoded by decoder.streaming_step
    # (carried conv buffers + KV cache, speech_tokenizer.py:882-930); their concatenation is the one-shot decode of the same codes
    forced = gc[None, :n].clone()
    it = model.generate(text, voice="vivian", lang_code="english", temperature=0.0, max_tokens=n, stream=True, streaming_interval=0.16, forced_codes=forced)
    first = next(it)
    assert first.is_streaming_chunk and first.token_count == 2 and first.samples == 2 * c["up"] and model.talker.frames_generated == 2 < n
    chunks = [first] + list(it)
    assert model.talker.frames_generated == n and sum(r.token_count for r in chunks) == n and chunks[-1].is_final_chunk and not chunks[0].is_final_chunk
    one_shot = model.speech_tokenizer.decoder(gc[:n].t()[None].contiguous())[0, 0].cpu()
    cat = torch.cat([r.audio for r in chunks]).cpu()
    assert cat.shape == one_shot.shape and float((cat - one_shot).abs().max()) <= 2e-5 * max(1.0, float(one_shot.abs().max()))   # fp32 rounding level (kernel choice by launch size)
    # ---- routing errors of the reference
    with pytest.raises(ValueError):
        list(model.generate(text, voice="nobody"))
    # this checkpoint has neither the speaker encoder nor the tokenizer's encoder half: like the reference (qwen3_tts.py:383, 1227-1231) a reference
    # clip then changes nothing (the cloning paths are covered by tests/test_qwen3_clone_gpu.py)
    assert model.speaker_encoder is None
    xa, tra, _ = model._prepare_generation_inputs(text, language="english", ref_audio=torch.zeros(100), ref_text="x")
    xb, trb, _ = model._prepare_generation_inputs(text, language="english")
    assert torch.equal(xa, xb) and torch.equal(tra, trb)
    assert not model.supports_tts_batch(ref_audio=torch.zeros(100), ref_text="x")


def test_qwen3_tts_batch_generate_left_padded(qwen3_ckpt):
    """batch_generate(): prompts of different lengths are left-padded (qwen3_tts.py:536-560); every sequence must reproduce its own
    single-sequence ORACLE run (no padding there), i.e. padding is invisible."""
    from mlx_audio_amd.tts.utils import load_model
    from mlx_audio_amd.tts.models.base import BatchGenerationResult
    from oracle.qwen3_talker_ref import Qwen3TalkerRef

    c = qwen3_ckpt
    model = load_model(c["path"], device=DEV)
    tok = FakeTokenizer(c["tc"].text_vocab_size)
    model.tokenizer = tok
    ref = Qwen3TalkerRef(c["tw"], c["tc"])
    texts = ["a short one", "a noticeably longer sentence than the first", "mid sized text"]
    speakers, instructs = [None, "vivian", None], [None, None, "speak slowly"]  # prompt lengths differ by the speaker slot / the instruct turn
    frames = 6
    bi = model._prepare_batch_inputs(texts, language="auto", speakers=speakers, instructs=instructs, return_metadata=True)
    assert bi.left_padding[2] == 0 and bi.left_padding[0] > bi.left_padding[1] > 0 and bi.input_embeds.shape[0] == 3
    assert bool((bi.input_embeds[0, : bi.left_padding[0]] == 0).all()) and bi.attention_mask.sum(1).tolist() == [float(n) for n in bi.prefill_lens]
    left = torch.tensor(bi.left_padding, dtype=torch.int32)
    out = model._frame_loop(bi.input_embeds, bi.trailing_text_hidden, bi.tts_pad_embed, frames, temperature=0.0, top_k=50, top_p=1.0,
                            repetition_penalty=1.05, left_pad=left, record=True)
    torch.cuda.synchronize()
    max_tr = bi.trailing_text_hidden.shape[1]
    for b, t in enumerate(texts):
        ex, etr, epad = _qwen3_ref_inputs(ref, c["tc"], tok, t, speaker=speakers[b], instruct=instructs[b])
        assert ex.shape[1] == bi.prefill_lens[b]
        etr = torch.cat([etr, epad.expand(1, max_tr - etr.shape[1], -1)], dim=1)  # right-padded with tts_pad like the batch (qwen3_tts.py:566-580)
        exp = ref.generate(ex, etr, epad, frames, temperature=0.0, record=True)
        # frame 0 logits: prefill through the left-padded batch == the unpadded single sequence
        e0, g0 = exp["trace"][0][0][0], out["trace"][0][0][b].cpu()
        assert float((g0 - e0).abs().max()) <= 2e-3 * float(e0.abs().max()), (b, float((g0 - e0).abs().max()))
        nf = min(exp["codes"].shape[1], out["codes"].shape[1])
        margins = [float(_gap(exp["trace"][f][i][0])) for f in range(nf) for i in range(exp["codes"].shape[2])]
        _margin.walk("qwen3_tts", out["codes"][b, :nf].cpu().flatten().tolist(), exp["codes"][0, :nf].flatten().tolist(), margins,
                     where=("batch_generate", b))
    res = list(model.batch_generate(texts, voices=speakers, instructs=instructs, temperature=0.0, max_tokens=frames))
    assert [r.sequence_idx for r in res] == [0, 1, 2] and all(isinstance(r, BatchGenerationResult) for r in res)
    assert all(r.samples == r.token_count * c["up"] == r.audio.shape[0] for r in res)


@pytest.mark.parametrize("max_batch", [3, 12])
def test_qwen3_tts_continuous_batching_session(qwen3_ckpt, max_batch):
    """``create_tts_batch_session`` (continuous_batching.py:37-360) on slot KV caches: requests arrive while others are mid-utterance, are admitted as
    slots free up, leave when they hit EOS / max_tokens -- and every request's codes equal ITS OWN single-sequence oracle run under the margin
    rule (sharing a step with sequences of other lengths, riding in a recycled slot, being prefilled next to other prompts: all invisible).
    max_batch 12: the steps run the rows pipeline (9..64 sequences per step)."""
    from mlx_audio_amd.tts.continuous import TTSBatchEvent, TTSBatchItem, TTSBatchOptions
    from mlx_audio_amd.tts.utils import load_model
    from oracle.qwen3_talker_ref import Qwen3TalkerRef

    c = qwen3_ckpt
    model = load_model(c["path"], device=DEV)
    tok = FakeTokenizer(c["tc"].text_vocab_size)
    model.tokenizer = tok
    ref = Qwen3TalkerRef(c["tw"], c["tc"])
    assert model.supports_tts_continuous_batch(voice=None) and not model.supports_tts_continuous_batch(ref_audio=torch.zeros(8))
    frames = 7
    texts = ["a short one", "a noticeably longer sentence than the first", "mid sized text", "four", "the fifth request arrives late and is long enough",
             "six", "seven is here", "eight", "nine nine nine", "ten", "eleven comes", "twelve", "thirteen is unlucky", "fourteen"]
    n_req = 7 if max_batch == 3 else len(texts)
    items = [TTSBatchItem(sequence_id=100 + i, text=texts[i], voice="vivian" if i % 3 == 1 else None) for i in range(n_req)]
    session = model.create_tts_batch_session(TTSBatchOptions(temperature=0.0, max_tokens=frames, max_batch_size=max_batch))
    session.trace, session.codes_log = [], {}
    assert session.idle and session.available_slots == max_batch
    arrivals = {0: items[:2], 2: items[2:5], 3: items[5:]} if max_batch == 3 else {0: items[:5], 1: items[5:11], 4: items[11:]}
    events, step = [], 0
    while step < 200 and (step <= max(arrivals) or not session.idle):
        if step in arrivals:
            session.add(arrivals[step])
        ev = session.step()
        assert len(session._active) <= max_batch and all(isinstance(e, TTSBatchEvent) and e.done for e in ev)
        events.extend(ev)
        step += 1
    torch.cuda.synchronize()
    assert session.idle and sorted(e.sequence_id for e in events) == [it.sequence_id for it in items]
    admits = [t for t in session.trace if t[0] == "admit"]
    assert admits[0][1] == [it.sequence_id for it in arrivals[0]][:max_batch]
    assert any(t[0] == "advance" and len(t[1]) > 1 for t in session.trace)
    if max_batch == 3:   # more requests than slots: later ones wait for a free slot and reuse it
        assert any(t[0] == "admit" and 1 <= len(t[1]) < 3 for t in session.trace[2:])
    for e in events:
        it = items[e.sequence_id - 100]
        ex, etr, epad = _qwen3_ref_inputs(ref, c["tc"], tok, it.text, speaker=it.voice)
        exp = ref.generate(ex, etr, epad, frames, temperature=0.0, record=True, pad_when_index_clamped=False)
        fa = int(exp["finished_at"][0])
        n_exp = fa if fa >= 0 else exp["codes"].shape[1]
        got = session.codes_log.get(e.sequence_id)
        gc = got.cpu() if got is not None else torch.zeros((0, exp["codes"].shape[2]), dtype=torch.long)
        nf = min(gc.shape[0], n_exp)
        margins = [float(_gap(exp["trace"][f][i][0])) for f in range(nf) for i in range(exp["codes"].shape[2])]
        done = _margin.walk("qwen3_tts", gc[:nf].flatten().tolist(), exp["codes"][0, :nf].flatten().tolist(), margins, where=("session", e.sequence_id))
        if done == nf * exp["codes"].shape[2]:   # no knife edge on the way: the lengths agree too
            assert e.token_count == n_exp, (e.sequence_id, e.token_count, n_exp)
        assert e.samples == e.token_count * c["up"] == int(e.audio.shape[0]) and e.sample_rate == 24000
    # cancel: a pending and an active request disappear without an event; max_tokens <= 0 answers with empty events
    s2 = model.create_tts_batch_session(TTSBatchOptions(temperature=0.0, max_tokens=50, max_batch_size=2))
    s2.add([TTSBatchItem(sequence_id=1, text="one"), TTSBatchItem(sequence_id=2, text="two"), TTSBatchItem(sequence_id=3, text="three")])
    s2.step()
    s2.cancel(1)
    s2.cancel(3)
    assert [st.sequence_id for st in s2._active] == [2] and not s2._pending
    s3 = model.create_tts_batch_session(TTSBatchOptions(max_tokens=0, max_batch_size=2))
    s3.add([TTSBatchItem(sequence_id=9, text="nothing")])
    ev = s3.step()
    assert len(ev) == 1 and ev[0].samples == 0 and ev[0].token_count == 0 and ev[0].done and s3.idle


# ------------------------------------------------------------------------------------------------ CSM
@pytest.fixture(scope="module")
def csm_ckpt(tmp_path_factory):
    from mlx_audio_amd.codec.models.mimi import mimi as M
    from mlx_audio_amd.tts.models.sesame import engine as E

    root = tmp_path_factory.mktemp("csm_tiny")
    cfg = E.tiny_csm()
    b, d = cfg.backbone, cfg.decoder
    mcfg = M.tiny_mimi_config()
    hf = lambda s: dict(hidden_size=s.d_model, num_hidden_layers=s.n_layers, num_attention_heads=s.n_heads, num_key_value_heads=s.n_kv_heads,
                        head_dim=s.head_dim, intermediate_size=s.d_ff, rms_norm_eps=s.norm_eps, rope_theta=s.rope_theta, max_position_embeddings=s.max_pos,
                        rope_scaling=dict(factor=s.rope_llama3_factor, rope_type="llama3"))
    config = dict(model_type="csm", **hf(b), depth_decoder_config=hf(d), audio_vocab_size=cfg.audio_vocab_size, audio_num_codebooks=cfg.audio_num_codebooks,
                  text_vocab_size=cfg.text_vocab_size, use_default_voice_prompt=False, audio_tokenizer_config=asdict(mcfg))
    (root / "config.json").write_text(json.dumps(config))
    w = E.make_csm_weights(cfg, seed=5)
    tt = {"wq": "attn.q_proj", "wk": "attn.k_proj", "wv": "attn.v_proj", "wo": "attn.output_proj", "w_gate": "mlp.w1", "w_down": "mlp.w2", "w_up": "mlp.w3"}
    ck = {}
    for k, v in w.items():  # torchtune names, as the sesame/csm-1b checkpoint stores them (sanitize renames them, sesame.py:577-604)
        m = re.match(r"^(backbone|decoder)\.layers\.(\d+)\.(\w+)\.weight$", k)
        if m and m.group(3) in tt:
            ck[f"{m.group(1)}.layers.{m.group(2)}.{tt[m.group(3)]}.weight"] = v
        elif m and m.group(3) in ("attn_norm", "mlp_norm"):
            ck[f"{m.group(1)}.layers.{m.group(2)}.{'sa_norm' if m.group(3) == 'attn_norm' else 'mlp_norm'}.scale"] = v
        elif k.endswith("final_norm.weight"):
            ck[k.replace("final_norm.weight", "norm.scale")] = v
        else:
            ck[k] = v
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in ck.items()}, str(root / "model.safetensors"))
    (root / "mimi").mkdir()
    mw = M.make_mimi_decoder_weights(mcfg, seed=6)
    save_file({k: v.contiguous() for k, v in mw.items()}, str(root / "mimi" / "model.safetensors"))
    return dict(path=root, cfg=cfg, w=w, mcfg=mcfg, mw=mw)


def test_csm_load_model_and_generate(csm_ckpt):
    from mlx_audio_amd.tts.utils import load_model
    from mlx_audio_amd.tts.models.base import GenerationResult
    from oracle import csm_ref as R
    from oracle.mimi_ref import MimiConfig as RMimiConfig, MimiDecoderRef

    c = csm_ckpt
    cfg = c["cfg"]
    model = load_model(c["path"], device=DEV)
    assert model.sample_rate == 24000 and model._audio_tokenizer is not None
    tok = FakeTokenizer(cfg.text_vocab_size)
    model._text_tokenizer = tok
    frames = 5
    text = "hi there"
    toks, mask = model._tokenize_text_segment(text, 0)
    ids = tok.encode("[0]" + text)
    assert toks.shape == (len(ids), cfg.audio_num_codebooks + 1) and toks[:, -1].tolist() == ids and bool(mask[:, -1].all()) and not bool(mask[:, :-1].any())
    rcfg = R.CSMConfig(backbone=R.StackConfig(**asdict(cfg.backbone)), decoder=R.StackConfig(**asdict(cfg.decoder)),
                       audio_vocab_size=cfg.audio_vocab_size, audio_num_codebooks=cfg.audio_num_codebooks, text_vocab_size=cfg.text_vocab_size)
    ref = R.CSMRef(c["w"], rcfg)
    exp = ref.generate(toks[None].long(), mask[None], frames, temperature=0.0, record=True)
    res = list(model.generate(text, speaker=0, temperature=0.0, max_audio_length_ms=frames * 80, unknown_cli_flag=1))
    assert len(res) == 1 and isinstance(res[0], GenerationResult) and res[0].sample_rate == 24000
    out = model.model.generate(toks[None], mask[None], frames, temperature=0.0)
    gf, ef = out["frames"][0].cpu(), exp["frames"][0]
    nf = min(gf.shape[0], ef.shape[0])
    margins = [float(_gap(exp["trace"][f][i][0])) for f in range(nf) for i in range(ef.shape[1])]
    checked = _margin.walk("csm", gf[:nf].flatten().tolist(), ef[:nf].flatten().tolist(), margins, where="Model.generate")
    assert checked >= 4, checked
    assert res[0].token_count == gf.shape[0] and res[0].samples == res[0].audio.shape[0] == gf.shape[0] * 1920
    import dataclasses

    names = {f.name for f in dataclasses.fields(RMimiConfig)}
    mref = MimiDecoderRef(c["mw"], RMimiConfig(**{k: v for k, v in asdict(c["mcfg"]).items() if k in names}))
    wav = mref(gf.t()[None].long())[0, 0]
    got = res[0].audio.cpu()
    assert got.shape == wav.shape and float((got - wav).abs().max()) <= 2e-3 * max(1.0, float(wav.abs().max()))
    # stream=True (sesame.py:825-860): chunks are decoded by the Mimi streaming decoder (carried state) and leave WHILE the frame loop runs -- the first
    # chunk (0.16 s = 2 frames) is in hand when only 2 of the 5 frames have been computed; the concatenation is the one-shot waveform
    it = model.generate(text, speaker=0, temperature=0.0, max_audio_length_ms=frames * 80, stream=True, streaming_interval=0.16)
    first = next(it)
    assert first.is_streaming_chunk and first.token_count == 2 and first.samples == 2 * 1920 and model.model.frames_generated == 2 < gf.shape[0]
    chunks = [first] + list(it)
    assert model.model.frames_generated == gf.shape[0] and [r.token_count for r in chunks] == [2, 2, 1][: len(chunks)] and sum(r.token_count for r in chunks) == gf.shape[0]
    cat = torch.cat([r.audio for r in chunks]).cpu()
    assert all(r.is_streaming_chunk for r in chunks) and cat.shape == got.shape and float((cat - got).abs().max()) <= 2e-5 * float(got.abs().max())   # fp32 rounding level
    # the 2048-position guard (sesame.py:817-820) and the audio-context refusals
    with pytest.raises(ValueError):
        list(model.generate("x" * 50, max_audio_length_ms=80 * 2040))
    with pytest.raises(NotImplementedError):
        list(model.generate(text, ref_audio=torch.zeros(100), ref_text="x"))
    # this checkpoint sets use_default_voice_prompt=False: like the reference (sesame.py:756) a voice name is then not looked at
    assert len(list(model.generate(text, voice="conversational_a", temperature=0.0, max_audio_length_ms=frames * 80))) == 1
    model._use_default_voice_prompt = True
    with pytest.raises(NotImplementedError):
        list(model.generate(text, voice="conversational_a"))


def test_csm_audio_context_through_the_mimi_encoder(csm_ckpt, tmp_path):
    """A Mimi checkpoint WITH the encoder half: ``_tokenize_audio`` (sesame.py:527-559) encodes the clip on the device (``codec.models.mimi.MimiEncoder``),
    its codes equal the oracle's ``Mimi.encode`` under the margin rule, and ``generate(ref_audio=..., ref_text=...)`` / ``context=[Segment(audio=...)]``
    run with the clip's frames in the prompt (the decode-only checkpoint of the test above refuses)."""
    import shutil

    from mlx_audio_amd.codec.models.mimi import mimi as M
    from mlx_audio_amd.tts.models.sesame.sesame import Segment
    from mlx_audio_amd.tts.utils import load_model
    from oracle.mimi_ref import MimiConfig as RMimiConfig, MimiEncoderRef

    c = csm_ckpt
    root = tmp_path / "csm_enc"
    shutil.copytree(c["path"], root)
    mw = {**c["mw"], **M.make_mimi_encoder_weights(c["mcfg"], seed=6)}
    save_file({k: v.contiguous() for k, v in mw.items()}, str(root / "mimi" / "model.safetensors"))
    model = load_model(root, device=DEV)
    model._text_tokenizer = FakeTokenizer(c["cfg"].text_vocab_size)
    K = c["cfg"].audio_num_codebooks
    clip = M.make_pcm(1, 1920 * 3 + 500, seed=4)[0, 0]
    frame, mask = model._tokenize_audio(clip.to(DEV))
    assert frame.shape == (5, K + 1) and mask.shape == frame.shape            # 4 frames (3 whole + 1 padded) + the all-zero EOS frame
    assert bool(mask[:, :K].all()) and not bool(mask[:, K].any()) and int(frame[-1].abs().sum()) == 0
    names = set(RMimiConfig.__dataclass_fields__)
    enc = MimiEncoderRef(mw, RMimiConfig(**{k: v for k, v in asdict(c["mcfg"]).items() if k in names}))
    want, margins = enc.quantize(enc.latent(clip[None, None]), return_margins=True)
    for t in range(4):
        _margin.walk("mimi_encode", frame[t, :1].tolist(), want[0, :1, t].tolist(), margins[0, :1, t].tolist(), thr=0.05, where=("csm ctx", t, 0))
        _margin.walk("mimi_encode", frame[t, 1:K].tolist(), want[0, 1:, t].tolist(), margins[0, 1:, t].tolist(), thr=0.05, where=("csm ctx", t))
    frames = 3
    a = list(model.generate("hi there", speaker=0, temperature=0.0, max_audio_length_ms=frames * 80, ref_audio=clip.to(DEV), ref_text="before"))
    b = list(model.generate("hi there", speaker=0, temperature=0.0, max_audio_length_ms=frames * 80,
                            context=[Segment(speaker=0, text="before", audio=clip.to(DEV))]))
    assert len(a) == len(b) == 1 and a[0].samples == a[0].audio.shape[0] > 0 and torch.equal(a[0].audio, b[0].audio)   # ref_audio IS the first segment (sesame.py:753-755)
