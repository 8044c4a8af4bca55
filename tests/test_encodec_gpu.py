"""EnCodec decode on MI355X against the CPU oracle and the reference run.  Needs a real MI355X: ``pytest -m gpu``."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def snr_db(got, want):
    got, want = got.double(), want.double()
    return float(10 * torch.log10((want ** 2).sum() / ((got - want) ** 2).sum().clamp_min(1e-300)))


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,T,H", [(1, 40, 64), (3, 75, 512), (12, 9, 128), (16, 33, 512), (40, 8, 64), (64, 5, 128)])
def test_lstm_seq_vs_float64(B, T, H):
    """``mi355_lstm_seq`` against float64 (x-projection given; the gates of the reference's Metal kernel, encodec.py:89-167), for the EnCodec width
    (512: Wh = 2 MB, the case the persistent one-CU LSTM cannot hold), odd and even step counts, 1 .. 64 sequences -- in BOTH forms: the one-launch step
    (``pack_lstm_seq_wh``: rows of Wh ordered by hidden unit, gates in the step GEMM's epilogue; round 5) and the two-launch step on block-ordered rows
    (GEMV kernels up to 8 sequences, the 9..64-row kernel above).  The two forms run the same GEMM arithmetic from 9 sequences on: equal to an ulp there."""
    from mlx_audio_amd import ops
    from oracle.encodec_ref import lstm_sigmoid

    ops.require_gpu()
    g = torch.Generator().manual_seed(B + T + H)
    wh = (torch.randn(4 * H, H, generator=g) / math.sqrt(H)).half().float()
    xp = torch.randn(B, T, 4 * H, generator=g)
    h, c, exp = torch.zeros(B, H, dtype=torch.float64), torch.zeros(B, H, dtype=torch.float64), []
    for t in range(T):
        gts = h @ wh.double().t() + xp[:, t].double()
        i, f, gg, o = lstm_sigmoid(gts[:, :H]), lstm_sigmoid(gts[:, H:2 * H]), torch.tanh(gts[:, 2 * H:3 * H]), lstm_sigmoid(gts[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        exp.append(h)
    exp = torch.stack(exp, 1)
    fused = ops.pack_lstm_seq_wh(wh, DEV, f16=True)
    assert fused.interleaved   # every H here is a multiple of 64
    res = {}
    for name, img in (("one launch per step", fused), ("two launches per step", ops.pack_rowmajor16(wh, None, DEV, f16=True))):
        out = torch.full((B, T, H), float("nan"), device=DEV)
        hT, cT = ops.lstm_seq(xp.to(DEV), img, out)
        torch.cuda.synchronize()
        assert rel_err(out, exp) < 2e-5 and rel_err(hT, h) < 2e-5 and rel_err(cT, c) < 2e-5, (name, rel_err(out, exp), rel_err(hT, h), rel_err(cT, c))
        res[name] = (out.clone(), hT.clone(), cT.clone())
    a, b = res["one launch per step"], res["two launches per step"]
    if B >= 9:   # same GEMM arithmetic; the cell update may contract differently in the two kernels (an ulp)
        assert all(rel_err(x, y) < 1e-6 for x, y in zip(a, b)), [rel_err(x, y) for x, y in zip(a, b)]
    # a second call on a carried state continues the sequence (h0 / c0 in, odd number of steps: the final state comes back from the scratch buffer)
    out2 = torch.empty((B, 3, H), device=DEV)
    x3 = torch.randn(B, 3, 4 * H, generator=g)
    h3, c3 = ops.lstm_seq(x3.to(DEV), fused, out2, h0=a[1], c0=a[2])
    hh, cc = h.clone(), c.clone()
    for t in range(3):
        gts = hh @ wh.double().t() + x3[:, t].double()
        i, f, gg, o = lstm_sigmoid(gts[:, :H]), lstm_sigmoid(gts[:, H:2 * H]), torch.tanh(gts[:, 2 * H:3 * H]), lstm_sigmoid(gts[:, 3 * H:])
        cc = f * cc + i * gg
        hh = o * torch.tanh(cc)
    assert rel_err(h3, hh) < 5e-5 and rel_err(c3, cc) < 5e-5 and rel_err(out2[:, -1], hh) < 5e-5


def test_encodec_engine_vs_reference_run():
    """The HIP decode path against what the reference's own ``Encodec.decode`` computed (tests/golden/ref_encodec_tiny.npz) and stage by stage against the oracle."""
    from mlx_audio_amd.codec.models import Encodec
    from mlx_audio_amd.codec.models.encodec.encodec import make_encodec_weights
    from oracle.encodec_ref import EncodecDecoderRef

    fx = np.load(os.path.join(GOLD, "ref_encodec_tiny.npz"))
    cfg = json.loads(str(fx["config"]))
    w = make_encodec_weights(cfg, seed=int(fx["seed_w"]))
    eng = Encodec(cfg, weights=w, device=DEV)
    codes = torch.from_numpy(fx["codes"]).long()
    got = eng.decode(codes, [None]).cpu()
    want = torch.from_numpy(fx["audio"])
    peak = float(want.abs().max())
    err = float((got - want).abs().max())
    print(f"encodec HIP vs reference run: max-abs {err:.2e} (peak {peak:.2f}), SNR {snr_db(got, want):.1f} dB")
    assert got.shape == want.shape and err <= 2e-3 * peak and snr_db(got, want) >= 50.0
    ref = EncodecDecoderRef(w, cfg)
    _, est = ref.decoder(ref.quantizer_decode(codes[:, 0]), return_stages=True)
    _, gst = eng._decoder(eng.quantizer.decode(codes[:, 0]), return_stages=True)
    torch.cuda.synchronize()
    for k, e in est.items():
        assert rel_err(gst[k], e) < 2e-3, (k, rel_err(gst[k], e))


def test_encodec_24khz_shapes_chunked_decode_and_batch():
    """The published 24 kHz model's sizes (32 filters, ratios 8/5/4/2, two 512-wide LSTM layers, 32 codebooks of 1024 x 128), a batch of 2, once as a
    single frame and once chunked (chunk_length_s / overlap: ``_linear_overlap_add`` of overlapping frames, encodec.py:652-677, 758-771), padding mask trim."""
    from mlx_audio_amd.codec.models import Encodec
    from mlx_audio_amd.codec.models.encodec.encodec import EncodecConfig, make_encodec_weights
    from oracle.encodec_ref import EncodecDecoderRef

    cfg = dict(upsampling_ratios=[8, 5, 4, 2], target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0])
    w = make_encodec_weights(cfg, seed=2)
    eng = Encodec(EncodecConfig(**cfg), weights=w, device=DEV)
    ref = EncodecDecoderRef(w, cfg)
    assert eng.quantizer.num_quantizers == 32 and eng.lstm[0]["H"] == 512 and eng.quantizer.get_num_quantizers_for_bandwidth(6.0) == 8
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, 1024, (2, 1, 8, 30), generator=g)   # 6 kbps: 8 codebooks, 30 frames = 0.4 s
    got = eng.decode(codes, [None], padding_mask=torch.ones(2, 9000)).cpu()
    want = ref.decode(codes, [None], padding_mask=torch.ones(2, 9000))
    assert got.shape == want.shape == (2, 9000, 1)
    err, peak = float((got - want).abs().max()), float(want.abs().max())
    print(f"encodec 24 kHz sizes: max-abs {err:.2e} (peak {peak:.2f}), SNR {snr_db(got, want):.1f} dB")
    assert err <= 2e-3 * peak and snr_db(got, want) >= 50.0
    # chunked: 3 chunks of 10 frames decoded separately, cross-faded with a triangular window at 50 % overlap
    ccfg = dict(cfg, chunk_length_s=3200 / 24000, overlap=0.5)
    eng_c, ref_c = Encodec(ccfg, weights=w, device=DEV), EncodecDecoderRef(w, ccfg)
    assert eng_c.chunk_length == 3200 and eng_c.chunk_stride == 1600
    chunks = torch.randint(0, 1024, (3, 2, 8, 10), generator=g)
    scales = [torch.full((2, 1, 1), 1.0 + 0.1 * i) for i in range(3)]
    got = eng_c.decode(chunks, scales).cpu()
    want = ref_c.decode(chunks, scales)
    assert got.shape == want.shape == (2, 2 * 1600 + 3200, 1)
    assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max()) and snr_db(got, want) >= 50.0
    with pytest.raises(ValueError, match="decode-only"):   # loaded without encoder weights (tests/test_codec_encode_gpu.py has the encode side)
        eng.encode(torch.zeros(1, 100, 1), bandwidth=eng.c["target_bandwidths"][0])
    with pytest.raises(ValueError):
        eng.decode(torch.zeros(2, 2, 8, 5, dtype=torch.long), [None])
