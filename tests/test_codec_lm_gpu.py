"""Qwen3-TTS talker / code predictor frame loop and CSM generate_frame (a28) on the HIP path vs the CPU
oracles.  Integer path = the sampled codes: compared bit-exactly under teacher forcing wherever the oracle's decision margin exceeds the
measured logit error (and free-running until the first knife-edge decision); logits <= 2e-3 of their peak.  Needs a real MI355X."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _margin  # noqa: E402  (tests/_margin.py: the knife-edge rule and its accounting)


def _gap(logits):
    top2 = torch.topk(logits, 2).values
    return top2[0] - top2[1]
DEV = "cuda"


def _gumbel(g, *shape):
    return -torch.log(-torch.log(torch.rand(*shape, generator=g).clamp_(1e-9, 1 - 1e-9)))


def _check_trace(exp_trace, got_trace, what):
    worst = 0.0
    for f, (ef, gf) in enumerate(zip(exp_trace, got_trace)):
        assert len(ef) == len(gf)
        for i, (e, g) in enumerate(zip(ef, gf)):
            err = float((g.cpu() - e).abs().max())
            peak = float(e.abs().max())
            assert err <= 2e-3 * peak, (what, f, i, err, peak)
            worst = max(worst, err / peak)
            top2 = torch.topk(e, 2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 10 * err
            if bool(clear.any()):
                assert torch.equal(g.cpu().argmax(-1)[clear], e.argmax(-1)[clear]), (what, f, i)
    return worst


@pytest.fixture(scope="module")
def talker():
    from mlx_audio_amd import ops
    from mlx_audio_amd.tts.models.qwen3_tts import talker as T
    from oracle.qwen3_talker_ref import Qwen3TalkerRef

    ops.require_gpu()
    cfg = T.tiny_talker_config()
    w = T.make_talker_weights(cfg, seed=1)
    g = torch.Generator().manual_seed(3)
    B, H = 3, cfg.hidden_size
    return dict(cfg=cfg, eng=T.Qwen3Talker(w, cfg, device=DEV), ref=Qwen3TalkerRef(w, cfg), B=B,
                pre=torch.randn(B, 7, H, generator=g) * 0.5, trail=torch.randn(B, 3, H, generator=g) * 0.5, pad=torch.randn(1, 1, H, generator=g) * 0.5)


def test_qwen3_text_projection(talker):
    ids = torch.randint(0, talker["cfg"].text_vocab_size, (2, 9))
    exp = talker["ref"].text_projection(talker["ref"].w["model.text_embedding.weight"][ids])
    got = talker["eng"].embed_text(ids)
    torch.cuda.synchronize()
    assert float((got.cpu() - exp).abs().max() / exp.abs().max()) < 2e-4


def test_qwen3_frame_loop_teacher_forced(talker):
    cfg, B = talker["cfg"], talker["B"]
    g = torch.Generator().manual_seed(5)
    frames = 6
    gu0, guc = _gumbel(g, frames, B, cfg.vocab_size), _gumbel(g, frames, cfg.num_code_groups - 1, B, cfg.code_predictor_config.vocab_size)
    kw = dict(temperature=0.9, top_k=50, top_p=0.95, repetition_penalty=1.05, gumbel0=gu0, gumbel_cp=guc)
    free = talker["ref"].generate(talker["pre"], talker["trail"], talker["pad"], frames, **kw)
    forced = free["codes"].clone()
    forced[1, 3, 0] = cfg.codec_eos_token_id   # sequence 1 ends at frame 3: finished rows keep emitting EOS, its history stops growing
    exp = talker["ref"].generate(talker["pre"], talker["trail"], talker["pad"], frames, forced_codes=forced, record=True, **kw)
    got = talker["eng"].generate(talker["pre"], talker["trail"], talker["pad"], frames, forced_codes=forced, record=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got["codes"].cpu(), exp["codes"])
    assert got["finished_at"].cpu().tolist() == exp["finished_at"].tolist() == [-1, 3, -1]
    _check_trace(exp["trace"], got["trace"], "qwen3")


def test_qwen3_frame_loop_teacher_forced_tall_batch(talker):
    """The same frame loop with 12 sequences per step: every single-position Linear (both stacks through the native runner's rows pipeline, the
    heads / projections through ``linear_rows``) runs mi355_rows_gemm / mi355_rows_finish, the fused attention step reads the projection slabs and
    writes planes.  Same oracle, same bars as the 3-sequence test."""
    cfg = talker["cfg"]
    g = torch.Generator().manual_seed(17)
    B, H, frames = 12, cfg.hidden_size, 4
    assert talker["eng"].talker.max_decode_rows == 64 and talker["eng"].cp.max_decode_rows == 64
    pre, trail = torch.randn(B, 7, H, generator=g) * 0.5, torch.randn(B, 3, H, generator=g) * 0.5
    gu0, guc = _gumbel(g, frames, B, cfg.vocab_size), _gumbel(g, frames, cfg.num_code_groups - 1, B, cfg.code_predictor_config.vocab_size)
    kw = dict(temperature=0.9, top_k=50, top_p=0.95, repetition_penalty=1.05, gumbel0=gu0, gumbel_cp=guc)
    free = talker["ref"].generate(pre, trail, talker["pad"], frames, **kw)
    forced = free["codes"].clone()
    forced[5, 2, 0] = cfg.codec_eos_token_id
    exp = talker["ref"].generate(pre, trail, talker["pad"], frames, forced_codes=forced, record=True, **kw)
    got = talker["eng"].generate(pre, trail, talker["pad"], frames, forced_codes=forced, record=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got["codes"].cpu(), exp["codes"])
    assert got["finished_at"].cpu().tolist() == exp["finished_at"].tolist()
    _check_trace(exp["trace"], got["trace"], "qwen3 (12 sequences)")


def test_qwen3_frame_loop_free_running_greedy(talker):
    frames = 5
    exp = talker["ref"].generate(talker["pre"], talker["trail"], talker["pad"], frames, temperature=0.0, record=True)
    ec = exp["codes"]
    nf, G = ec.shape[1], ec.shape[2]
    # the oracle's token is forced at its knife-edge decisions (top-2 gap < _margin.THR) and nowhere else: the engine free-runs between them on a
    # re-synchronised context, so EVERY other decision of every sequence is compared (tests/_margin.py: walk_resync)
    margins = torch.tensor([[[float(_gap(exp["trace"][f][i][b])) for i in range(G)] for f in range(nf)] for b in range(ec.shape[0])])
    forced = torch.where(margins < _margin.THR, ec[:, :nf], torch.full_like(ec[:, :nf], -1))
    got = talker["eng"].generate(talker["pre"], talker["trail"], talker["pad"], nf, temperature=0.0, forced_codes=forced)
    torch.cuda.synchronize()
    gc = got["codes"].cpu()
    assert gc.shape[1] == nf
    for b in range(ec.shape[0]):
        _margin.walk_resync("qwen3_tts", gc[b].flatten().tolist(), ec[b].flatten().tolist(), margins[b].flatten().tolist(), where=("talker", b))
    # ... and the unforced loop (the EOS poll and the stop rule) reproduces the oracle wherever no knife edge precedes
    free = talker["eng"].generate(talker["pre"], talker["trail"], talker["pad"], frames, temperature=0.0, poll=2)
    fc = free["codes"].cpu()
    for b in range(ec.shape[0]):
        n = min(nf, fc.shape[1]) * G
        ke = _margin.knife_edges(margins[b].flatten().tolist())
        stop = ke[0] if ke else n
        assert fc[b].flatten().tolist()[:min(stop, n)] == ec[b].flatten().tolist()[:min(stop, n)]


@pytest.fixture(scope="module")
def csm():
    from mlx_audio_amd import ops
    from mlx_audio_amd.tts.models.sesame import engine as E
    from oracle import csm_ref as R

    ops.require_gpu()
    cfg = E.tiny_csm()
    w = E.make_csm_weights(cfg, seed=2)
    rcfg = R.CSMConfig(backbone=R.StackConfig(**asdict(cfg.backbone)), decoder=R.StackConfig(**asdict(cfg.decoder)),
                       audio_vocab_size=cfg.audio_vocab_size, audio_num_codebooks=cfg.audio_num_codebooks, text_vocab_size=cfg.text_vocab_size)
    g = torch.Generator().manual_seed(4)
    B, S, nb = 2, 9, cfg.audio_num_codebooks
    toks = torch.zeros(B, S, nb + 1, dtype=torch.long)
    mask = torch.zeros(B, S, nb + 1, dtype=torch.bool)
    toks[:, :5, -1] = torch.randint(0, cfg.text_vocab_size, (B, 5), generator=g)
    mask[:, :5, -1] = True
    toks[:, 5:, :nb] = torch.randint(1, cfg.audio_vocab_size, (B, 4, nb), generator=g)
    mask[:, 5:, :nb] = True
    return dict(cfg=cfg, eng=E.CSMEngine(w, cfg, device=DEV), ref=R.CSMRef(w, rcfg), toks=toks, mask=mask, B=B)


def test_csm_frames_teacher_forced(csm):
    cfg, B = csm["cfg"], csm["B"]
    g = torch.Generator().manual_seed(6)
    frames = 4
    gum = _gumbel(g, frames, cfg.audio_num_codebooks, B, cfg.audio_vocab_size)
    free = csm["ref"].generate(csm["toks"], csm["mask"], frames, gumbel=gum)
    forced = free["frames"]
    assert forced.shape[1] == frames
    exp = csm["ref"].generate(csm["toks"], csm["mask"], frames, gumbel=gum, forced=forced, record=True)
    got = csm["eng"].generate(csm["toks"], csm["mask"], frames, gumbel=gum, forced=forced, record=True)
    torch.cuda.synchronize()
    assert torch.equal(got["frames"].cpu(), exp["frames"])
    _check_trace(exp["trace"], got["trace"], "csm")


def test_csm_free_running_greedy_and_eos(csm):
    exp = csm["ref"].generate(csm["toks"], csm["mask"], 3, temperature=0.0, record=True)
    ef = exp["frames"]
    margins = torch.tensor([[[float(_gap(exp["trace"][f][i][b])) for i in range(ef.shape[2])] for f in range(ef.shape[1])] for b in range(ef.shape[0])])
    forced = torch.where(margins < _margin.THR, ef, torch.full_like(ef, -1))   # the oracle's token at its knife edges only (tests/_margin.py)
    got = csm["eng"].generate(csm["toks"], csm["mask"], ef.shape[1], temperature=0.0, forced=forced)
    torch.cuda.synchronize()
    gf = got["frames"].cpu()
    assert gf.shape == ef.shape
    for b in range(ef.shape[0]):
        _margin.walk_resync("csm", gf[b].flatten().tolist(), ef[b].flatten().tolist(), margins[b].flatten().tolist(), where=("csm", b))
    free = csm["eng"].generate(csm["toks"], csm["mask"], 3, temperature=0.0, poll=1)["frames"].cpu()   # the unforced loop, up to the first knife edge
    for b in range(ef.shape[0]):
        ke = _margin.knife_edges(margins[b].flatten().tolist())
        n = min(free.shape[1], ef.shape[1]) * ef.shape[2]
        stop = min(ke[0] if ke else n, n)
        assert free[b].flatten().tolist()[:stop] == ef[b].flatten().tolist()[:stop]
    # EOS: an all-zero frame stops the loop and is not returned (sesame.py:828)
    forced = torch.zeros(csm["B"], 2, csm["cfg"].audio_num_codebooks, dtype=torch.long)
    forced[:, 0] = 5
    out = csm["eng"].generate(csm["toks"], csm["mask"], 2, forced=forced)
    assert out["frames"].shape[1] == 1
