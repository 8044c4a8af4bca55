"""Parity of the transformer-side HIP kernels (called through the C ABI) against fp64 torch / the CPU oracle.

  * mi355_flash_attention  (f32 MFMA flash kernel + KV-streaming decode kernel): GQA, causal, window, ragged lengths
  * mi355_gemv             (decode-step skinny GEMM on row-major 16-bit weights)
  * mi355_conv_gemm        new modes: precision 4 (fp16 hi+lo), ELU / SnakeBeta prologues, SiLU / GELU-tanh / ELU / tanh
                           epilogues, per-column scale (LayerScale)
  * mi355_whisper_greedy_step / mi355_softmax_prob_at vs the restated logit filters (bit-exact tokens and masks)

Needs a real MI355X: ``pytest -m gpu``.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from mlx_audio_amd import ops as _ops

    _ops.require_gpu()
    return _ops


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, heads, kv_heads, dh, scale, causal, window, lens_q, lens_k):
    """fp64 reference with the visibility rule of mi355_flash_attn_args.  q [B,Tq,H*dh], k/v [B,Tk,G*dh]."""
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    out = torch.zeros(B, Tq, heads * dh, dtype=torch.float64)
    grp = heads // kv_heads
    for b in range(B):
        lq = int(lens_q[b]) if lens_q is not None else Tq
        lk = int(lens_k[b]) if lens_k is not None else Tk
        for h in range(heads):
            g = h // grp
            qq = q[b, :lq, h * dh:(h + 1) * dh].double()
            kk = k[b, :lk, g * dh:(g + 1) * dh].double()
            vv = v[b, :lk, g * dh:(g + 1) * dh].double()
            s = (qq @ kk.T) * scale
            i = torch.arange(lq)[:, None] + (lk - lq)
            j = torch.arange(lk)[None, :]
            vis = torch.ones(lq, lk, dtype=torch.bool)
            if causal:
                vis &= j <= i
            if window > 0:
                vis &= j > i - window
            s = s.masked_fill(~vis, -float("inf"))
            p = torch.softmax(s, dim=-1)
            p = torch.nan_to_num(p, nan=0.0)  # rows without a visible key -> zeros
            out[b, :lq, h * dh:(h + 1) * dh] = p @ vv
    return out


ATT_CASES = [
    # B, Tq, Tk, heads, kv_heads, dh, causal, window, ragged, mode
    (2, 200, 200, 3, 3, 64, False, 0, False, 1),
    (1, 1500, 1500, 2, 2, 64, False, 0, False, 1),     # the Whisper encoder shape (2 of 12 heads)
    (2, 130, 130, 4, 2, 64, True, 0, True, 1),         # GQA + causal + ragged
    (1, 97, 300, 2, 1, 128, True, 0, False, 1),        # dh 128, queries = last 97 of 300 keys
    (2, 70, 70, 2, 2, 128, False, 0, True, 1),
    (1, 300, 300, 2, 2, 64, True, 50, False, 1),       # sliding window (Mimi context)
    (3, 1, 77, 4, 2, 64, True, 0, True, 2),            # decode step
    (2, 1, 1500, 12, 12, 64, False, 0, False, 2),      # Whisper cross-attention decode step
    (2, 3, 3, 2, 2, 64, True, 0, False, 2),            # Whisper prefill (3 initial tokens)
    (1, 2, 40, 16, 8, 128, True, 0, False, 2),         # Qwen3 code predictor step 0 (2 positions), dh 128
    (2, 1, 400, 8, 8, 64, True, 250, False, 2),        # Mimi decode step with context 250
    (2, 5, 33, 2, 2, 64, True, 0, True, 0),            # auto -> decode kernel
    (1, 9, 33, 2, 2, 64, True, 0, False, 0),           # auto -> flash kernel
    (1, 40, 40, 2, 1, 64, True, 0, False, 2),          # decode kernel forced on a longer block
    (2, 1, 700, 4, 2, 128, True, 0, True, 2),          # 16-wave decode kernel, dh 128, ragged
    (1, 2, 257, 2, 2, 64, True, 0, False, 2),          # just past the 4 -> 16 wave switch
]


@pytest.mark.parametrize("B,Tq,Tk,heads,kvh,dh,causal,window,ragged,mode", ATT_CASES)
def test_flash_attention(ops, B, Tq, Tk, heads, kvh, dh, causal, window, ragged, mode):
    g = torch.Generator().manual_seed(Tq * 7 + Tk + heads)
    ldq, ldk = heads * dh + 8, 2 * kvh * dh  # strided views: q padded, k|v interleaved in one buffer like a KV cache
    qb = torch.randn(B, Tq, ldq, generator=g)
    kvb = torch.randn(B, Tk, ldk, generator=g)
    q = qb[:, :, :heads * dh]
    k, v = kvb[:, :, :kvh * dh], kvb[:, :, kvh * dh:]
    lens_q = lens_k = None
    if ragged:
        lens_k = torch.tensor([max(1, Tk - 13 * i) for i in range(B)], dtype=torch.int32)
        lens_q = torch.tensor([min(Tq, int(lens_k[i])) - (i % 2) * min(2, Tq - 1) for i in range(B)], dtype=torch.int32)
    scale = 1.0 / math.sqrt(dh)
    exp = ref_attention(q, k, v, heads, kvh, dh, scale, causal, window, lens_q, lens_k)
    qd, kvd = qb.to(DEV), kvb.to(DEV)
    out = torch.full((B, Tq, heads * dh), 7.0, device=DEV)
    ops.flash_attention(qd[:, :, :heads * dh], kvd[:, :, :kvh * dh], kvd[:, :, kvh * dh:], out, heads=heads, kv_heads=kvh, dh=dh,
                        scale=scale, causal=causal, window=window, lens_q=None if lens_q is None else lens_q.to(DEV),
                        lens_k=None if lens_k is None else lens_k.to(DEV), mode=mode)
    torch.cuda.synchronize()
    got = out.cpu()
    for b in range(B):
        lq = int(lens_q[b]) if lens_q is not None else Tq
        assert rel_err(got[b, :lq], exp[b, :lq]) < 2e-5, (b, rel_err(got[b, :lq], exp[b, :lq]))
        if lq < Tq:
            assert torch.all(got[b, lq:] == 7.0), "rows past lens_q must not be written"


@pytest.mark.parametrize("Tq,mode", [(1, 2), (70, 1)])
def test_attention_left_padded_batch(ops, Tq, mode):
    """BatchKVCache layout (lm/models/cache.py:502-560): rows are left-padded, k_start[b] keys of padding are invisible."""
    g = torch.Generator().manual_seed(17 + Tq)
    B, Tk, H, dh = 3, 90, 2, 64
    pad = [0, 13, 41]
    q = torch.randn(B, Tq, H * dh, generator=g)
    k = torch.randn(B, Tk, H * dh, generator=g)
    v = torch.randn(B, Tk, H * dh, generator=g)
    out = torch.empty(B, Tq, H * dh, device=DEV)
    ops.flash_attention(q.to(DEV), k.to(DEV), v.to(DEV), out, heads=H, dh=dh, causal=True, mode=mode,
                        k_start=torch.tensor(pad, dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    for b in range(B):
        p = pad[b]
        nq = min(Tq, Tk - p)  # queries that fall inside the padding have no meaning in the reference either
        exp = ref_attention(q[b:b + 1, Tq - nq:], k[b:b + 1, p:], v[b:b + 1, p:], H, H, dh, 1.0 / math.sqrt(dh), True, 0, None, None)
        assert rel_err(out[b, Tq - nq:].cpu(), exp[0]) < 2e-5, b


@pytest.mark.parametrize("Tq,mode", [(1, 2), (3, 2), (40, 1)])
def test_attention_head_major_kv(ops, Tq, mode):
    """K / V as head-major planes [B, kv_heads, Tk, dh] (Whisper's cross-attention layout) == the packed-row layout."""
    g = torch.Generator().manual_seed(23 + Tq)
    B, Tk, H, G, dh = 2, 333, 4, 2, 64
    q = torch.randn(B, Tq, H * dh, generator=g)
    k = torch.randn(B, Tk, G * dh, generator=g)
    v = torch.randn(B, Tk, G * dh, generator=g)
    exp = ref_attention(q, k, v, H, G, dh, 1.0 / math.sqrt(dh), False, 0, None, None)
    kh = k.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous().to(DEV)
    vh = v.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous().to(DEV)
    out = torch.empty(B, Tq, H * dh, device=DEV)
    ops.flash_attention(q.to(DEV), kh, vh, out, heads=H, kv_heads=G, dh=dh, mode=mode, head_major=True)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), exp) < 2e-5


@pytest.mark.parametrize("kvd", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Tq,Tk,dh,G,hm,causal", [(1, 1500, 64, 4, True, False), (1, 31, 128, 2, False, True), (3, 333, 64, 2, False, True),
                                                   (1, 700, 128, 4, True, False), (40, 333, 64, 2, False, True), (150, 260, 128, 1, True, False)])
def test_attention_16bit_kv(ops, kvd, Tq, Tk, dh, G, hm, causal):
    """K / V held in the checkpoint's 16-bit type (mi355_flash_attn_args.kv_dtype; the reference caches them in the model dtype, whisper.py:360-361,
    lm/models/cache.py:104-176): decode kernel (4 and 16 waves, both head sizes, packed rows and head-major planes) and prefill kernel against
    float64 attention over the SAME rounded K / V -- the kernels only change how the bytes are read, so the fp32 bar stays."""
    g = torch.Generator().manual_seed(101 + Tq + Tk + dh)
    B, H = 2, 4
    q = torch.randn(B, Tq, H * dh, generator=g)
    k = torch.randn(B, Tk, G * dh, generator=g).to(kvd)
    v = torch.randn(B, Tk, G * dh, generator=g).to(kvd)
    exp = ref_attention(q, k.float(), v.float(), H, G, dh, 1.0 / math.sqrt(dh), causal, 0, None, None)
    if hm:
        kd = k.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous().to(DEV)
        vd = v.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous().to(DEV)
    else:  # a strided "cache": rows of k | v side by side, capacity beyond Tk
        cache = torch.zeros(B, Tk + 5, 2 * G * dh, dtype=kvd, device=DEV)
        cache[:, :Tk, : G * dh] = k.to(DEV)
        cache[:, :Tk, G * dh:] = v.to(DEV)
        kd, vd = cache[:, :Tk, : G * dh], cache[:, :Tk, G * dh:]
    out = torch.empty(B, Tq, H * dh, device=DEV)
    ops.flash_attention(q.to(DEV), kd, vd, out, heads=H, kv_heads=G, dh=dh, causal=causal, head_major=hm)
    torch.cuda.synchronize()
    # prefill (Tq > 8) runs both contractions on the 16-bit matrix pipe with Q and P split hi + lo: ~16 mantissa bits for bf16 (the conv path's bar)
    assert rel_err(out.cpu(), exp) < (3e-5 if (Tq > 8 and kvd == torch.bfloat16) else 2e-5), rel_err(out.cpu(), exp)


@pytest.mark.parametrize("dh,H,G,Tk,mode,norms,kvd", [(128, 16, 8, 37, 0, True, torch.float32), (128, 8, 2, 1, 0, True, torch.float32),
                                                      (64, 8, 8, 300, 1, False, torch.float32), (128, 4, 2, 700, 0, True, torch.bfloat16),
                                                      (64, 4, 1, 65, 0, True, torch.float16)])
def test_decode_attention_with_fused_norm_rope_and_cache_store(ops, dh, H, G, Tk, mode, norms, kvd):
    """The fused decode step (mi355_flash_attn_args.new_k): q / k per-head RMSNorm + rotary embedding + cache store + attention in one launch ==
    head_norm_rope (q in place, k into the cache slot) followed by the plain decode attention, for both rope modes, with and without the norms,
    left padding, float32 and 16-bit caches.  The cache row written by the fused kernel must equal the one the separate kernel writes."""
    g = torch.Generator().manual_seed(7 + dh + Tk)
    B = 3
    nq, nkv = H * dh, 2 * G * dh
    q_raw = torch.randn(B, nq, generator=g).to(DEV)
    kv_raw = torch.randn(B, nkv, generator=g).to(DEV)
    cache0 = torch.randn(B, Tk + 3, nkv, generator=g).to(kvd).to(DEV)
    k_start = torch.tensor([0, min(2, Tk - 1), 0], dtype=torch.int32, device=DEV)
    qw = (1.0 + 0.1 * torch.randn(dh, generator=g)).to(DEV) if norms else None
    kw = (1.0 + 0.1 * torch.randn(dh, generator=g)).to(DEV) if norms else None
    pos = Tk - 1
    inv = 1.0 / (10000.0 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    ang = torch.arange(Tk + 4, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    # ---- separate kernels: norm + rope of q (in place) and of k (float32 scratch), v untouched, then the store and the attention
    q_ref = q_raw.clone()[:, None, :]
    kv_ref = kv_raw.clone()[:, None, :]
    ops.head_norm_rope(q_ref, q_ref, heads=H, dh=dh, norm_weight=qw, eps=1e-6, cos=cos, sin=sin, pos0=pos, interleaved=(mode == 1), pos_sub=k_start,
                       second=(kv_ref[:, :, : G * dh], kv_ref[:, :, : G * dh], G, kw))
    cache_ref = cache0.clone()
    cache_ref[:, Tk - 1] = kv_ref[:, 0].to(kvd)
    out_ref = torch.empty(B, 1, nq, device=DEV)
    ops.flash_attention(q_ref, cache_ref[:, :Tk, : G * dh], cache_ref[:, :Tk, G * dh:], out_ref, heads=H, kv_heads=G, dh=dh, causal=True, k_start=k_start, mode=2)
    # ---- fused
    cache = cache0.clone()
    out = torch.empty(B, 1, nq, device=DEV)
    ops.flash_attention(q_raw[:, None, :], cache[:, :Tk, : G * dh], cache[:, :Tk, G * dh:], out, heads=H, kv_heads=G, dh=dh, causal=True, k_start=k_start, mode=2,
                        fused=dict(new_k=kv_raw[:, : G * dh], new_v=kv_raw[:, G * dh:], q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos=cos, sin=sin,
                                   rope_mode=mode, pos=pos))
    torch.cuda.synchronize()
    assert torch.equal(cache[:, Tk - 1], cache_ref[:, Tk - 1])            # same arithmetic, same rounding into the cache dtype
    assert torch.equal(cache[:, :Tk - 1], cache0[:, :Tk - 1]) and torch.equal(cache[:, Tk:], cache0[:, Tk:])   # nothing else touched
    assert rel_err(out.cpu(), out_ref.cpu()) < 2e-5, rel_err(out.cpu(), out_ref.cpu())   # (the new k | v enter the softmax as the cache will hold them)


@pytest.mark.parametrize("Tq,Tk,nsplit,causal,hm", [(1, 1500, 2, False, True), (1, 1500, 4, False, False), (3, 700, 8, True, False), (1, 130, 3, True, False),
                                                      (2, 64, 8, True, False)])
def test_attention_key_split_decode(ops, Tq, Tk, nsplit, causal, hm):
    """Flash-decoding: the key range split over several workgroups, partials merged by the last one; repeated launches reuse the tickets."""
    g = torch.Generator().manual_seed(31 + Tk + nsplit)
    B, H, G, dh = 2, 4, 2, 64
    q = torch.randn(B, Tq, H * dh, generator=g)
    k = torch.randn(B, Tk, G * dh, generator=g)
    v = torch.randn(B, Tk, G * dh, generator=g)
    exp = ref_attention(q, k, v, H, G, dh, 1.0 / math.sqrt(dh), causal, 0, None, None)
    kd, vd = k.to(DEV), v.to(DEV)
    if hm:
        kd = kd.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous()
        vd = vd.reshape(B, Tk, G, dh).permute(0, 2, 1, 3).contiguous()
    for rep in range(3):
        out = torch.full((B, Tq, H * dh), 5.0, device=DEV)
        ops.flash_attention(q.to(DEV), kd, vd, out, heads=H, kv_heads=G, dh=dh, causal=causal, mode=2, head_major=hm, nsplit=nsplit)
        torch.cuda.synchronize()
        assert rel_err(out.cpu(), exp) < 2e-5, rep
    ws, cnt = ops.attn_split_workspace(torch.device(DEV), B * H * Tq, dh)
    assert int(cnt.abs().sum()) == 0  # tickets are left zeroed


def test_flash_and_decode_kernels_agree(ops):
    """Same inputs through both kernels (mode 1 / mode 2): a cross-check that does not involve the reference at all."""
    g = torch.Generator().manual_seed(5)
    B, T, H, dh = 2, 64, 4, 64
    x = torch.randn(B, T, 3 * H * dh, generator=g).to(DEV)
    o1 = torch.empty(B, T, H * dh, device=DEV)
    o2 = torch.empty_like(o1)
    for o, mode in ((o1, 1), (o2, 2)):
        ops.flash_attention(x[:, :, :H * dh], x[:, :, H * dh:2 * H * dh], x[:, :, 2 * H * dh:], o, heads=H, dh=dh, causal=True, mode=mode)
    torch.cuda.synchronize()
    assert rel_err(o1.cpu(), o2.cpu()) < 1e-5


# ------------------------------------------------------------------------------------------------ gemv
def _round16(t, f16):
    return t.to(torch.float16 if f16 else torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,f16,act,use_res,use_cs", [
    (1, 768, 768, True, 0, False, False),
    (1, 51865, 768, True, 0, False, False),     # Whisper logits (ragged N, NC = 4)
    (2, 3072, 768, True, 3, False, False),      # mlp1 + GELU
    (3, 768, 3072, True, 0, True, False),       # mlp2 + residual, K > one LDS chunk
    (5, 2048, 2048, False, 0, True, True),
    (8, 1024, 6144, False, 5, False, False),    # SiLU, 8 rows
    (4, 30, 1024, False, 6, False, False),      # tiny N (NC = 1 tail), GELU-tanh
    (1, 2051, 2048, False, 0, False, False),    # CSM audio head (odd N)
    (1, 1024, 8192, False, 0, True, True),      # CSM depth-decoder down projection: one row, K > 2048 -> split-K kernel, residual + LayerScale
    (1, 2050, 3072, True, 3, True, False),      # split K, fp16 weights, N not a multiple of 4, GELU
    (1, 16384, 1024, False, 0, False, False),   # one row, wide N: two columns per wave, grid-stride column groups
    (1, 1536, 1024, False, 0, True, False),     # depth-decoder q|k|v shape (register-resident x, one column per wave)
    (1, 40, 2048, False, 5, False, False),      # fewer column groups than waves
])
def test_gemv(ops, M, N, K, f16, act, use_res, use_cs):
    g = torch.Generator().manual_seed(M * 100 + N + K)
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), f16)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K + 4, generator=g)[:, :K]  # strided rows
    res = torch.randn(M, N, generator=g) if use_res else None
    cs = torch.randn(N, generator=g) if use_cs else None
    v = x.double() @ w.double().T + bias.double()
    if act == 3:
        v = F.gelu(v)
    elif act == 5:
        v = F.silu(v)
    elif act == 6:
        v = F.gelu(v, approximate="tanh")
    if cs is not None:
        v = v * cs.double()
    if res is not None:
        v = v + res.double()
    rw = ops.pack_rowmajor16(w, bias, DEV, f16=f16)
    xd = torch.zeros(M, K + 4, device=DEV)
    xd[:, :K] = x.to(DEV)
    y = torch.empty(M, N, device=DEV)
    ops.gemv(xd[:, :K], rw, y, post_act=act, res=None if res is None else res.to(DEV), colscale=None if cs is None else cs.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), v) < _gemv_tol(M, K, f16), rel_err(y.cpu(), v)


@pytest.mark.parametrize("N,K,idx,off", [(1024, 2048, 7, 100), (512, 4096, 0, 3)])
def test_gemv_gathered_input_row(ops, N, K, idx, off):
    """mi355_gemv_args.x_ids: the input row is row (ids[0] + offset) of a table, read on the device (embedding lookup fused into the projection,
    sesame.py:392-396) == the GEMV on that row."""
    g = torch.Generator().manual_seed(N + K)
    table = torch.randn(200, K, generator=g).to(DEV)
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), False)
    rw = ops.pack_rowmajor16(w, None, DEV)
    ids = torch.tensor([idx], dtype=torch.int32, device=DEV)
    y = torch.empty(1, N, device=DEV)
    ops.gemv(table, rw, y, x_ids=ids, x_id_offset=off)
    y2 = torch.empty(1, N, device=DEV)
    ops.gemv(table[idx + off: idx + off + 1], rw, y2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert rel_err(y.cpu(), table[idx + off].cpu().double() @ w.double().T) < 5e-6


def _gemv_tol(M, K, f16, base=5e-6):
    """5..8 rows with K % 64 == 0 run on the matrix pipe (gemv_mfma.hip): the input rows are split into hi + lo images of the weights' type,
    ~16 mantissa bits for bf16 weights (the split conv_gemm precision 2 runs prefill with; its tests allow 3e-5), ~22 for fp16."""
    return 2e-5 if (5 <= M <= 8 and K % 64 == 0 and not f16) else base


@pytest.mark.parametrize("mode,M,K", [("layer", 1, 768), ("layer", 5, 3072), ("rms", 8, 2048), ("rms", 2, 1024), ("rms", 1, 2048), ("rms", 1, 1024),
                                      ("layer", 1, 2048), ("layer", 1, 3072)])
def test_gemv_fused_norm_and_split(ops, mode, M, K):
    g = torch.Generator().manual_seed(K + M)
    Nq, Nkv = 256, 128
    w = _round16(torch.randn(Nq + Nkv, K, generator=g) / math.sqrt(K), False)
    bias = torch.randn(Nq + Nkv, generator=g) * 0.1
    x = torch.randn(M, K, generator=g) * 2.0 + 0.3
    nw, nb = torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.1
    xd = x.double()
    if mode == "layer":
        xn = F.layer_norm(xd, (K,), nw.double(), nb.double(), 1e-5)
    else:
        nb = None
        xn = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * nw.double()
    exp = xn @ w.double().T + bias.double()
    rw = ops.pack_rowmajor16(w, bias, DEV)
    y = torch.empty(M, Nq, device=DEV)
    cache = torch.zeros(M, 7, Nkv + 8, device=DEV)          # a strided "KV-cache slot" destination
    ops.gemv(x.to(DEV), rw, y, norm=(mode, nw.to(DEV), None if nb is None else nb.to(DEV), 1e-5 if mode == "layer" else 1e-6),
             y2=cache[:, 3, :Nkv])
    torch.cuda.synchronize()
    tol = _gemv_tol(M, K, False, base=1e-5)
    assert rel_err(y.cpu(), exp[:, :Nq]) < tol, rel_err(y.cpu(), exp[:, :Nq])
    assert rel_err(cache[:, 3, :Nkv].cpu(), exp[:, Nq:]) < tol
    assert float(cache[:, 2].abs().max()) == 0.0 and float(cache[:, 3, Nkv:].abs().max()) == 0.0


@pytest.mark.parametrize("M", [3, 1])
def test_gemv_swiglu(ops, M):
    g = torch.Generator().manual_seed(11)
    I, K = 3072, 1024
    wg = _round16(torch.randn(I, K, generator=g) / math.sqrt(K), False)
    wu = _round16(torch.randn(I, K, generator=g) / math.sqrt(K), False)
    x = torch.randn(M, K, generator=g)
    exp = F.silu(x.double() @ wg.double().T) * (x.double() @ wu.double().T)
    inter = torch.stack([wg, wu], dim=1).reshape(2 * I, K)  # gate_0, up_0, gate_1, up_1, ...
    rw = ops.pack_rowmajor16(inter, None, DEV, f16=False)
    y = torch.empty(M, I, device=DEV)
    ops.gemv(x.to(DEV), rw, y, glu=True)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), exp) < 5e-6


# ------------------------------------------------------------------------------------------------ conv_gemm additions
def _ref_conv(x, w, b, dil, pad):
    k = w.shape[1]
    xp = F.pad(x.transpose(1, 2).double(), (pad, (k - 1) * dil - pad))
    return F.conv1d(xp, w.permute(0, 2, 1).double(), None if b is None else b.double(), dilation=dil).transpose(1, 2)


@pytest.mark.parametrize("cin,cout,k,dil,L,B,tile", [
    (768, 768, 1, 1, 300, 2, 0),
    (80, 768, 3, 1, 300, 1, 0),        # Whisper conv1 (C_in not a multiple of 32)
    (1536, 768, 2, 1, 150, 2, 0),      # Whisper conv2 as the 2-tap pair conv
    (768, 3072, 1, 1, 1500, 1, 0),
    (128, 128, 7, 1, 1100, 3, 6128128),
    (768, 3072, 1, 1, 1500, 1, 6128128),
    (96, 51865, 1, 1, 5, 2, 0),        # logits-shaped: huge N, few rows
])
def test_conv_gemm_precision4(ops, cin, cout, k, dil, L, B, tile):
    g = torch.Generator().manual_seed(cin + cout + k)
    w = (torch.randn(cout, k, cin, generator=g) / math.sqrt(cin * k)).to(torch.float16).to(torch.float32)
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, L, cin, generator=g) * 3.0
    pad = (k - 1) * dil // 2
    exp = _ref_conv(x, w, bias, dil, pad)
    pc = ops.pack_conv(w, bias, DEV, f16=True)
    y = torch.empty(B, L, cout, device=DEV)
    ops.conv_gemm(x.to(DEV), pc, y, dil=dil, pad=pad, precision=4, tile=tile)
    y3 = torch.empty(B, L, cout, device=DEV)
    ops.conv_gemm(x.to(DEV), pc, y3, dil=dil, pad=pad, precision=3, tile=tile)
    torch.cuda.synchronize()
    e4, e3 = rel_err(y.cpu(), exp), rel_err(y3.cpu(), exp)
    assert e4 < 3e-6, e4            # fp16 hi+lo: ~22 mantissa bits of the activation
    assert e3 < 2e-3 and e4 < e3    # single fp16 pass: the reference's own activation rounding


@pytest.mark.parametrize("tile", [0, 6128128])
def test_conv_gemm_new_activations(ops, tile):
    """ELU / SnakeBeta prologues, SiLU / GELU-tanh / ELU / tanh epilogues and the per-column scale, on the 4-wave kernels (auto) and on the
    per-family instantiations of the wave-specialised kernel (tile code 6128128)."""
    g = torch.Generator().manual_seed(77)
    B, L, cin, cout, k = 2, 333, 96, 160, 7
    w = (torch.randn(cout, k, cin, generator=g) / math.sqrt(cin * k)).to(torch.bfloat16).to(torch.float32)
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, L, cin, generator=g)
    res = torch.randn(B, L, cout, generator=g)
    cs = torch.randn(cout, generator=g)
    alpha = torch.rand(cin, generator=g) + 0.5
    inv_beta = 1.0 / (torch.rand(cin, generator=g) + 0.5)
    pc = ops.pack_conv(w, bias, DEV)
    xd, resd = x.to(DEV), res.to(DEV)
    pad = k - 1  # causal

    def run(**kw):
        y = torch.empty(B, L, cout, device=DEV)
        ops.conv_gemm(xd, pc, y, pad=pad, tile=tile, **kw)
        torch.cuda.synchronize()
        return y.cpu()

    xe = F.elu(x.double())
    exp = _ref_conv(xe, w, bias, 1, pad)
    assert rel_err(run(pre_act=ops.ACT_ELU), exp) < 5e-5
    al = torch.zeros(128); al[:cin] = alpha
    ib = torch.zeros(128); ib[:cin] = inv_beta
    xs = x.double() + inv_beta.double() * torch.sin(alpha.double() * x.double()) ** 2
    exp = _ref_conv(xs, w, bias, 1, pad)
    assert rel_err(run(pre_act=ops.ACT_SNAKE, pre_alpha=al.to(DEV), pre_inv_beta=ib.to(DEV)), exp) < 5e-5
    base = _ref_conv(x, w, bias, 1, pad)
    assert rel_err(run(colscale=cs.to(DEV), res=resd), base * cs.double() + res.double()) < 5e-5
    if tile:
        # the wave-specialised kernel carries one epilogue activation per instantiation, SiLU and GELU-tanh for the linear layers (K = 1:
        # GEMM mode) that use them (text projection, Mimi / codec transformer MLPs); ELU / tanh epilogues only exist on the 4-wave kernels
        w1 = (torch.randn(cout, 1, cin, generator=g) / math.sqrt(cin)).to(torch.bfloat16).to(torch.float32)
        pc1 = ops.pack_conv(w1, bias, DEV)
        base1 = _ref_conv(x, w1, bias, 1, 0)
        for act, fn in ((ops.ACT_SILU, F.silu), (ops.ACT_GELU_TANH, lambda t: F.gelu(t, approximate="tanh")), (ops.ACT_GELU, F.gelu)):
            y = torch.empty(B, L, cout, device=DEV)
            ops.conv_gemm(xd, pc1, y, tile=tile, post_act=act, colscale=cs.to(DEV), res=resd)
            torch.cuda.synchronize()
            assert rel_err(y.cpu(), fn(base1) * cs.double() + res.double()) < 5e-5
        with pytest.raises(Exception):
            run(post_act=ops.ACT_TANH)
        return
    assert rel_err(run(post_act=ops.ACT_SILU), F.silu(base)) < 5e-5
    assert rel_err(run(post_act=ops.ACT_GELU_TANH), F.gelu(base, approximate="tanh")) < 5e-5
    assert rel_err(run(post_act=ops.ACT_ELU), F.elu(base)) < 5e-5
    assert rel_err(run(post_act=ops.ACT_TANH), torch.tanh(base)) < 5e-5
    assert rel_err(run(colscale=cs.to(DEV), res=resd), base * cs.double() + res.double()) < 5e-5


# ------------------------------------------------------------------------------------------------ whisper decode step
def _tok():
    from oracle.whisper_ref import TokenizerSpec

    return TokenizerSpec()


@pytest.mark.parametrize("case", ["first", "after_text", "after_ts_pair", "after_single_ts", "ended", "no_ts_rules"])
@pytest.mark.parametrize("split", [False, True])
def test_whisper_greedy_step_matches_filters(ops, case, split):
    """``split``: the row spread over 16 workgroups (two launches, the last workgroup of a row merges): same filtered logits, tokens, log-probs."""
    from oracle import whisper_ref as R

    tok = _tok()
    V, B = 51865, 3
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    logits = torch.randn(B, V, generator=g) * 3.0
    logits[0, tok.timestamp_begin + 7] += 25.0   # a sequence where timestamps dominate
    logits[1, 1234] += 25.0                      # a sequence where text dominates
    init = list(tok.sot_sequence)
    sb = len(init)
    hist = {"first": [], "after_text": [tok.timestamp_begin, 400, 500], "after_ts_pair": [tok.timestamp_begin, 400, tok.timestamp_begin + 50, tok.timestamp_begin + 50],
            "after_single_ts": [tok.timestamp_begin, 400, tok.timestamp_begin + 50], "ended": [tok.timestamp_begin, tok.eot], "no_ts_rules": [400, 401]}[case]
    tokens = torch.tensor([init + hist] * B, dtype=torch.long)
    if case == "ended":
        tokens[1, -1] = 777  # only sequences 0 and 2 have ended
    ts_rules = case != "no_ts_rules"
    suppress = [1, 2, 3, tok.sot, tok.no_speech, 1234 + 1]
    filters = [R.SuppressBlankRef(tok, sb, V), R.SuppressTokensRef(suppress, V)]
    if ts_rules:
        filters.append(R.ApplyTimestampRulesRef(tok, sb, 50))
    lg = logits.clone()
    for f in filters:
        lg = f.apply(lg, tokens)
    sum0 = torch.tensor([-1.0, -2.0, -3.0])
    exp_tokens, _, exp_sum = R.GreedyDecoderRef(tok.eot).update(tokens, lg, sum0.clone())

    n = tokens.shape[1]
    tk = torch.full((B, n + 4), -5, dtype=torch.int32)
    tk[:, :n] = tokens.to(torch.int32)
    tkd = tk.to(DEV)
    sums = sum0.to(DEV)
    smask = torch.zeros(V); smask[suppress] = -float("inf")
    ld = V + 3
    lgd = torch.zeros(B, ld, device=DEV)
    lgd[:, :V] = logits.to(DEV)
    filt = torch.zeros(B, ld, device=DEV)
    blank = torch.tensor(list(tok.blank_ids) + [tok.eot], dtype=torch.int32, device=DEV)
    ops.whisper_greedy_step(lgd, tkd, n, sb, sums, V=V, suppress_mask=smask.to(DEV), blank_ids=blank, timestamp_rules=ts_rules,
                            timestamp_begin=tok.timestamp_begin, eot=tok.eot, no_timestamps=tok.no_timestamps, max_initial_timestamp_index=50,
                            filtered=filt, split_ws=ops.whisper_step_workspace(DEV, B) if split else None)
    torch.cuda.synchronize()
    got_f = filt[:, :V].cpu()
    assert torch.equal(torch.isinf(got_f), torch.isinf(lg)), "mask pattern differs"
    fin = torch.isfinite(lg)
    assert torch.equal(got_f[fin], lg[fin]), "filters only add 0 / -inf: finite logits must be bit-identical"
    assert torch.equal(tkd[:, :n + 1].cpu().long(), exp_tokens)
    assert torch.all(tkd[:, n + 1:].cpu() == -5)
    np.testing.assert_allclose(sums.cpu().numpy(), exp_sum.numpy(), rtol=2e-5, atol=2e-5)


def test_whisper_step_forced_and_no_speech(ops):
    tok = _tok()
    V, B = 51865, 2
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B, V, generator=g)
    lgd = logits.to(DEV)
    p = ops.softmax_prob_at(lgd, tok.no_speech)
    exp = torch.softmax(logits.double(), -1)[:, tok.no_speech]
    np.testing.assert_allclose(p.cpu().numpy(), exp.numpy(), rtol=1e-5)
    init = list(tok.sot_sequence_including_notimestamps)
    tk = torch.zeros(B, 16, dtype=torch.int32)
    tk[:, :len(init)] = torch.tensor(init, dtype=torch.int32)
    tkd = tk.to(DEV)
    sums = torch.zeros(B, device=DEV)
    forced = torch.tensor([4321, 99], dtype=torch.int32, device=DEV)
    ops.whisper_greedy_step(lgd, tkd, len(init), len(init), sums, eot=tok.eot, forced_next=forced)
    torch.cuda.synchronize()
    assert tkd[:, len(init)].cpu().tolist() == [4321, 99]
    lp = torch.log_softmax(logits.double(), -1)
    np.testing.assert_allclose(sums.cpu().numpy(), [float(lp[0, 4321]), float(lp[1, 99])], rtol=1e-5)


@pytest.mark.parametrize("M,N,K,act,use_res,glu", [(1, 1000, 1024, 0, False, False), (8, 514, 2048, 3, True, False), (3, 256, 3072, 0, True, False),
                                                   (1, 514, 4096, 0, True, False), (1, 4096, 2048, 0, False, True), (5, 1000, 1024, 5, False, False),
                                                   (7, 64, 64, 0, True, False), (6, 2050, 1536, 0, False, True), (8, 16384, 1024, 0, False, True),
                                                   (5, 8200, 2048, 5, True, False), (8, 1024, 8192, 0, True, False), (6, 520, 4160, 3, False, False),
                                                   (5, 2048, 1040, 0, False, True), (2, 130, 16, 5, False, False), (8, 6144, 2048, 0, False, True)])
def test_gemv_fp8_weights(ops, M, N, K, act, use_res, glu):
    """mi355_gemv on an fp8 (OCP e4m3fn, power-of-two row scales) image against float64 on the dequantised weights the oracle restates
    (oracle/lm_ref.py quantize_rows_fp8_ref): the kernel's byte decode and scale fold are exact, so the bar is the bf16 path's."""
    from oracle.lm_ref import dequantize_rows_fp8_ref, quantize_rows_fp8_ref

    g = torch.Generator().manual_seed(M * 1000 + N + K)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    w[1] = 0.0
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K + 4, generator=g)[:, :K]
    res = torch.randn(M, N, generator=g) if use_res else None
    wq = dequantize_rows_fp8_ref(*quantize_rows_fp8_ref(w)).double()
    v = x.double() @ wq.T + bias.double()
    if glu:
        v = F.silu(v[:, 0::2]) * v[:, 1::2]
    elif act == 3:
        v = F.gelu(v)
    elif act == 5:
        v = F.silu(v)
    if res is not None:
        v = v + res.double()
    rw, wdq = ops.pack_rowmajor_fp8(w, bias, DEV)
    assert rw.wdtype == 2 and torch.equal(wdq.double(), wq)
    xd = torch.zeros(M, K + 4, device=DEV)
    xd[:, :K] = x.to(DEV)
    y = torch.empty(M, N // 2 if glu else N, device=DEV)
    ops.gemv(xd[:, :K], rw, y, post_act=act, res=None if res is None else res.to(DEV), glu=glu)
    torch.cuda.synchronize()
    # 5..8 rows with K % 64 == 0 run on the fp8 matrix pipe (1, 2 or 4 column tiles per workgroup, 2048-column chunks of K): the input rows are split into four e4m3 terms (~16 significant bits, the
    # accuracy of the bf16 hi + lo split: same bar as the 16-bit matrix-pipe kernel)
    # (four e4m3 terms = ~16 significant bits of the input rows; a SwiGLU output multiplies two such results: 3e-5, the conv path's hi + lo bar)
    tol = 3e-5 if (5 <= M <= 8 and K % 64 == 0) else 5e-6
    assert rel_err(y.cpu(), v) < tol, rel_err(y.cpu(), v)


@pytest.mark.parametrize("M,N,K,f16,mode,glu,act,use_res,split", [
    (8, 2304, 768, True, "layer", False, 0, False, 768),      # Whisper q | k | v: pre-LN fused, k | v into a strided cache slot
    (8, 51865, 768, True, "layer", False, 0, False, 0),       # Whisper logits: ragged last tile
    (8, 768, 3072, True, None, False, 0, True, 0),            # Whisper mlp2 + residual, two k chunks
    (8, 3072, 768, True, "layer", False, 3, False, 0),        # mlp1 + GELU
    (8, 12288, 2048, False, "rms", True, 0, False, 0),        # talker gate | up with fused RMSNorm + SwiGLU
    (6, 2048, 6144, False, None, False, 0, True, 0),          # talker down + residual, three k chunks, 6 rows
    (5, 1040, 1024, False, "rms", False, 5, False, 0),        # N not a multiple of 16, SiLU
    (7, 30, 64, False, None, False, 0, False, 0),             # one k step, tiny N
    # ---- 9..64 rows: gemm_rows.hip (a batch of sequences per decode step; BASELINE config[3] = 64 utterances)
    (64, 4096, 2048, False, "rms", False, 0, False, 2048),    # talker q | k | v at 64 rows: fused RMSNorm, k | v into a strided cache slot
    (64, 12288, 2048, False, "rms", True, 0, False, 0),       # talker gate | up: fused RMSNorm + SwiGLU, 768 tiles over 512 workgroups
    (64, 2048, 6144, False, None, False, 0, True, 0),         # talker down + residual, 24 chunks of K
    (64, 2048, 2048, False, None, False, 0, True, 0),         # talker o-proj + residual
    (33, 1024, 3072, False, None, False, 5, True, 0),         # code predictor down, 33 rows (3 row groups of the 64-row kernel), SiLU
    (32, 6144, 1024, False, "rms", True, 0, False, 0),        # code predictor gate | up at 32 rows (the 32-row kernel)
    (17, 3072, 2048, False, "rms", False, 0, False, 0),       # codec head with the deferred final norm, 17 rows
    (16, 2051, 2048, False, None, False, 0, False, 0),        # 16 rows (the 16-row kernel), N not a multiple of 16
    (9, 1040, 1024, False, "rms", False, 5, False, 0),        # 9 rows, ragged last tile
    (12, 51865, 768, True, "layer", False, 0, False, 0),      # Whisper logits at 12 rows: LayerNorm with bias, fp16 weights, 7 tile groups
    (24, 2304, 768, True, "layer", False, 0, False, 768),     # Whisper q | k | v at 24 rows
    (40, 768, 3072, True, None, False, 3, True, 0),           # Whisper mlp2 at 40 rows, GELU + residual
    (64, 48, 64, False, None, False, 0, False, 0),            # one k step, 3 tiles
    (10, 30, 128, False, "layer", False, 0, False, 0),        # tiny
])
def test_gemv_matrix_pipe(ops, M, N, K, f16, mode, glu, act, use_res, split):
    """gemv_mfma.hip (5..8 rows) and gemm_rows.hip (9..64 rows): every epilogue / prologue combination the decode steps use, against float64."""
    g = torch.Generator().manual_seed(M + N + K)
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), f16)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K + 8, generator=g)[:, :K] * 1.5 + 0.2
    res = torch.randn(M, N, generator=g) if use_res else None
    xd = x.double()
    nw = nb = None
    if mode == "layer":
        nw, nb = torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.1
        xd = F.layer_norm(xd, (K,), nw.double(), nb.double(), 1e-5)
    elif mode == "rms":
        nw = torch.randn(K, generator=g)
        xd = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * nw.double()
    v = xd @ w.double().T + bias.double()
    if glu:
        v = F.silu(v[:, 0::2]) * v[:, 1::2]
    elif act == 3:
        v = F.gelu(v)
    elif act == 5:
        v = F.silu(v)
    if res is not None:
        v = v + res.double()
    rw = ops.pack_rowmajor16(w, bias, DEV, f16=f16)
    xdev = torch.zeros(M, K + 8, device=DEV)
    xdev[:, :K] = x.to(DEV)
    n_out = N // 2 if glu else (split if split else N)
    y = torch.full((M, n_out), float("nan"), device=DEV)
    cache = torch.zeros(M, 3, (N - split) + 8, device=DEV) if split else None
    norm = None if mode is None else (mode, nw.to(DEV), None if nb is None else nb.to(DEV), 1e-5 if mode == "layer" else 1e-6)
    ops.gemv(xdev[:, :K], rw, y, post_act=act, res=None if res is None else res.to(DEV), glu=glu, norm=norm,
             y2=None if not split else cache[:, 1, : N - split])
    torch.cuda.synchronize()
    tol = 3e-6 if f16 else 2e-5
    if split:
        assert rel_err(y.cpu(), v[:, :split]) < tol, rel_err(y.cpu(), v[:, :split])
        assert rel_err(cache[:, 1, : N - split].cpu(), v[:, split:]) < tol
        assert float(cache[:, 0].abs().max()) == 0.0 and float(cache[:, 2].abs().max()) == 0.0 and float(cache[:, 1, N - split:].abs().max()) == 0.0
    else:
        assert torch.isfinite(y).all()
        assert rel_err(y.cpu(), v) < tol, rel_err(y.cpu(), v)


@pytest.mark.parametrize("M,heads,kv_heads,dh,K,fp8", [(1, 4, 1, 64, 256, False), (3, 8, 2, 128, 1024, False), (1, 32, 8, 64, 2048, False), (2, 4, 2, 64, 512, True)])
def test_gemv_fused_interleaved_rope(ops, M, heads, kv_heads, dh, K, fp8):
    """q | k | v projection with the interleaved rotary embedding (nn.RoPE(traditional=True), sesame/attention.py:41-105) applied in the GEMV
    epilogue: q and k columns rotated at one position, v untouched, k | v landing in a strided cache slot; against float64."""
    g = torch.Generator().manual_seed(M + heads + K)
    nq, nk = heads * dh, kv_heads * dh
    N = nq + 2 * nk
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), False)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K, generator=g)
    nw = torch.randn(K, generator=g)
    pos = 37
    inv = 1.0 / (10000.0 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    ang = pos * inv
    cos_row, sin_row = torch.cos(ang), torch.sin(ang)
    if fp8:
        from oracle.lm_ref import dequantize_rows_fp8_ref, quantize_rows_fp8_ref

        rw, _ = ops.pack_rowmajor_fp8(w, bias, DEV)
        wd = dequantize_rows_fp8_ref(*quantize_rows_fp8_ref(w)).double()
    else:
        rw = ops.pack_rowmajor16(w, bias, DEV)
        wd = w.double()
    xd = x.double()
    xn = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.double()
    v = xn @ wd.T + bias.double()
    exp = v.clone()
    qk = v[:, : nq + nk].reshape(M, heads + kv_heads, dh // 2, 2)
    c, s_ = cos_row.double(), sin_row.double()
    rot = torch.stack([qk[..., 0] * c - qk[..., 1] * s_, qk[..., 1] * c + qk[..., 0] * s_], dim=-1)
    exp[:, : nq + nk] = rot.reshape(M, nq + nk)
    y = torch.empty(M, nq, device=DEV)
    cache = torch.zeros(M, 5, 2 * nk + 8, device=DEV)
    ops.gemv(x.to(DEV), rw, y, norm=("rms", nw.to(DEV), None, 1e-5), y2=cache[:, 2, : 2 * nk],
             rope=(cos_row.to(DEV), sin_row.to(DEV), dh, nq + nk))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), exp[:, :nq]) < 1e-5, rel_err(y.cpu(), exp[:, :nq])
    assert rel_err(cache[:, 2, : 2 * nk].cpu(), exp[:, nq:]) < 1e-5
    assert float(cache[:, 1].abs().max()) == 0.0 and float(cache[:, 3].abs().max()) == 0.0 and float(cache[:, 2, 2 * nk:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ rows pipeline (9..64 sequences per decode step)
def test_tile_image_matches_host_packer(ops):
    """The device-side permutation that builds tile images == mi355_pack_tiles16_host, element for element (ragged last tile, both 16-bit types)."""
    g = torch.Generator().manual_seed(3)
    for n, k, f16 in [(37, 128, False), (64, 64, True), (2051, 192, False)]:
        w = _round16(torch.randn(n, k, generator=g), f16)
        t = ops.tiles16_from_rowmajor(ops.pack_rowmajor16(w, None, DEV, f16=f16))
        assert np.array_equal(t.w.cpu().numpy().view(np.uint16), ops.pack_tiles16_host(w, f16=f16))


@pytest.mark.parametrize("M,N,K,f16,mode,glu,act,use_res,split,kg", [
    (64, 4096, 2048, False, "rms", False, 0, False, 2048, None),   # talker q | k | v: RMSNorm in the converter, k | v into a strided cache slot
    (64, 12288, 2048, False, "rms", True, 0, False, 0, None),      # talker gate | up + SwiGLU
    (64, 2048, 6144, False, None, False, 0, True, 0, None),        # talker down + residual (10 K groups)
    (64, 2048, 2048, False, None, False, 0, True, 0, None),        # talker o-proj + residual
    (33, 1024, 3072, False, None, False, 5, True, 0, None),        # code predictor down at 33 rows (R = 64), SiLU
    (32, 6144, 1024, False, "rms", True, 0, False, 0, None),       # code predictor gate | up at 32 rows (R = 32)
    (17, 3072, 2048, False, "rms", False, 0, False, 0, None),      # codec head (R = 32)
    (16, 2051, 2048, False, None, False, 0, False, 0, None),       # R = 16, ragged last tile (T = 2)
    (9, 1040, 1024, False, "rms", False, 5, False, 0, 1),          # one K group
    (12, 51865, 768, True, "layer", False, 0, False, 0, None),     # Whisper logits: LayerNorm with bias, fp16, 3242 tiles
    (24, 2304, 768, True, "layer", False, 0, False, 768, 2),       # Whisper q | k | v, two K groups
    (40, 768, 3072, True, None, False, 3, True, 0, None),          # Whisper mlp2, GELU + residual
    (64, 48, 64, False, None, False, 0, False, 0, None),           # one k step, T = 1
    (10, 30, 128, False, "layer", False, 0, False, 0, None),       # tiny
])
def test_rows_pipeline(ops, M, N, K, f16, mode, glu, act, use_res, split, kg):
    """fp32 rows -> (norm) -> planes [mi355_rows_finish as converter] -> mi355_rows_gemm -> slabs -> mi355_rows_finish (epilogue), against float64."""
    g = torch.Generator().manual_seed(M + N + K)
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), f16)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K + 8, generator=g)[:, :K] * 1.5 + 0.2
    res = torch.randn(M, N, generator=g) if use_res else None
    xd = x.double()
    nw = nb = None
    if mode == "layer":
        nw, nb = torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.1
        xd = F.layer_norm(xd, (K,), nw.double(), nb.double(), 1e-5)
    elif mode == "rms":
        nw = torch.randn(K, generator=g)
        xd = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * nw.double()
    v = xd @ w.double().T + bias.double()
    if glu:
        v = F.silu(v[:, 0::2]) * v[:, 1::2]
    elif act == 3:
        v = F.gelu(v)
    elif act == 5:
        v = F.silu(v)
    if res is not None:
        v = v + res.double()
    tl = ops.tiles16_from_rowmajor(ops.pack_rowmajor16(w, None, DEV, f16=f16))
    R = ops.rows_R(M)
    xdev = torch.zeros(M, K + 8, device=DEV)
    xdev[:, :K] = x.to(DEV)
    planes = ops.rows_planes(R, K, DEV)
    planes.fill_(0x7fc0 if not f16 else 0x7e00)   # NaN in every row: rows >= M must never reach a stored result
    norm = None if mode is None else (mode, nw.to(DEV), None if nb is None else nb.to(DEV), 1e-5 if mode == "layer" else 1e-6)
    ops.rows_finish(xdev[:, :K], M, K, norm=norm, planes=planes, R=R, f16=f16)
    kgroups = ops.rows_kgroups(N, K) if kg is None else kg
    ld = ops.round_up(N, 8)
    part = torch.full((kgroups, M, ld), float("nan"), device=DEV)
    assert ops.rows_gemm(planes, tl, part, M, R, kgroups=kgroups) == kgroups
    n_out = N // 2 if glu else (split if split else N)
    y = torch.full((M, n_out), float("nan"), device=DEV)
    cache = torch.zeros(M, 3, (N - split) + 8, device=DEV) if split else None
    ops.rows_finish(part, M, N, kgroups, bias=bias.to(DEV), post_act=act, res=None if res is None else res.to(DEV), glu=glu, y=y,
                    y2=None if not split else cache[:, 1, : N - split])
    torch.cuda.synchronize()
    tol = 3e-6 if f16 else 2e-5
    if split:
        assert rel_err(y.cpu(), v[:, :split]) < tol, rel_err(y.cpu(), v[:, :split])
        assert rel_err(cache[:, 1, : N - split].cpu(), v[:, split:]) < tol
        assert float(cache[:, 0].abs().max()) == 0.0 and float(cache[:, 2].abs().max()) == 0.0 and float(cache[:, 1, N - split:].abs().max()) == 0.0
    else:
        assert torch.isfinite(y).all()
        assert rel_err(y.cpu(), v) < tol, rel_err(y.cpu(), v)


@pytest.mark.parametrize("M,mode,kvd", [(64, "rms", torch.bfloat16), (20, "layer", torch.float16)])
def test_rows_finish_outputs(ops, M, mode, kvd):
    """The row epilogue's other destinations: in-place residual (y aliases res), LayerScale, the normalised row as fp32 (yn) and as planes feeding a
    second GEMM (checked through that GEMM), and a 16-bit KV-cache slot."""
    g = torch.Generator().manual_seed(M)
    K, N, N2 = 256, 512, 192
    w1 = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), False)
    w2 = _round16(torch.randn(N2, N, generator=g) / math.sqrt(N), False)
    x, res, cs = torch.randn(M, K, generator=g), torch.randn(M, N, generator=g), torch.randn(N, generator=g)
    nw, nb = torch.randn(N, generator=g), (torch.randn(N, generator=g) * 0.1 if mode == "layer" else None)
    h = (x.double() @ w1.double().T) * cs.double() + res.double()
    hn = F.layer_norm(h, (N,), nw.double(), nb.double(), 1e-5) if mode == "layer" else h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.double()
    out = hn @ w2.double().T
    R = ops.rows_R(M)
    t1 = ops.tiles16_from_rowmajor(ops.pack_rowmajor16(w1, None, DEV))
    t2 = ops.tiles16_from_rowmajor(ops.pack_rowmajor16(w2, None, DEV))
    p1, p2 = ops.rows_planes(R, K, DEV), ops.rows_planes(R, N, DEV)
    ops.rows_finish(x.to(DEV), M, K, planes=p1, R=R)
    kg1 = ops.rows_kgroups(N, K)
    part = torch.empty(kg1, M, N, device=DEV)
    ops.rows_gemm(p1, t1, part, M, R)
    stream = res.to(DEV).clone()   # the residual stream, updated in place
    yn = torch.empty(M, N, device=DEV)
    ops.rows_finish(part, M, N, kg1, colscale=cs.to(DEV), res=stream, y=stream, norm=(mode, nw.to(DEV), None if nb is None else nb.to(DEV), 1e-5), yn=yn,
                    planes=p2, R=R)
    kg2 = ops.rows_kgroups(N2, N)
    part2 = torch.empty(kg2, M, N2, device=DEV)
    ops.rows_gemm(p2, t2, part2, M, R)
    y = torch.empty(M, 64, device=DEV)
    slot = torch.zeros(M, 2, N2 - 64 + 8, dtype=kvd, device=DEV)
    ops.rows_finish(part2, M, N2, kg2, y=y, y2=slot[:, 1, : N2 - 64])
    torch.cuda.synchronize()
    assert rel_err(stream.cpu(), h) < 2e-5 and rel_err(yn.cpu(), hn) < 3e-5
    assert rel_err(y.cpu(), out[:, :64]) < 4e-5
    exp_slot = out[:, 64:].to(torch.float32).to(kvd).double()
    assert rel_err(slot[:, 1, : N2 - 64].double().cpu(), exp_slot) < 1e-2 and float(slot[:, 0].abs().max()) == 0.0


def _decode_planes(planes, R, K, f16):
    """hi + lo planes (fragment order, see mi355audio.h) -> float64 [R, K]."""
    raw = planes[: 2 * R * K].cpu().view(torch.float16 if f16 else torch.bfloat16).double().reshape(K // 64, 2, 2, 4, R, 8)
    x = raw[:, 0] + raw[:, 1]                      # [step, half, group, row, 8]
    return x.permute(3, 0, 2, 1, 4).reshape(R, K)  # column = 64 step + 16 group + 8 half + e


@pytest.mark.parametrize("M,N,K,use_bias", [(64, 12288, 2048, False), (20, 512, 128, True), (9, 256, 64, True)])
def test_rows_gemm_fused_swiglu_epilogue(ops, M, N, K, use_bias):
    """One K group: mi355_rows_gemm applies SwiGLU itself and writes the result as planes (no row-epilogue launch); against float64, and the planes
    written by the converter decode back to the input (the layout the header documents)."""
    g = torch.Generator().manual_seed(M + N)
    w = _round16(torch.randn(N, K, generator=g) / math.sqrt(K), False)
    bias = torch.randn(N, generator=g) * 0.1 if use_bias else None
    x = torch.randn(M, K, generator=g)
    v = x.double() @ w.double().T + (bias.double() if use_bias else 0.0)
    exp = F.silu(v[:, 0::2]) * v[:, 1::2]
    R = ops.rows_R(M)
    tl = ops.tiles16_from_rowmajor(ops.pack_rowmajor16(w, None, DEV))
    pin, pout = ops.rows_planes(R, K, DEV), ops.rows_planes(R, N // 2, DEV)
    ops.rows_finish(x.to(DEV), M, K, planes=pin, R=R)
    ops.rows_gemm(pin, tl, None, M, R, kgroups=1, glu_planes_out=pout, glu_bias=None if bias is None else bias.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(_decode_planes(pin, R, K, False)[:M], x.double()) < 1e-5
    assert rel_err(_decode_planes(pout, R, N // 2, False)[:M], exp) < 3e-5


@pytest.mark.parametrize("M,N,K,glu,fused", [(8, 4096, 1024, False, False), (64, 2048, 2048, False, False), (8, 16384, 1024, True, True), (20, 512, 256, True, False)])
def test_rows_pipeline_fp8_tile_image(ops, M, N, K, glu, fused):
    """fp8 tile images (BASELINE config[4]: fp8 GEMMs): the e4m3 bytes of ``mi355_pack_rowmajor_fp8_host`` in tile order are decoded to bf16 in registers
    (exact) and multiplied on the bf16 matrix pipe with hi + lo input planes; the per-row power-of-two scales are applied to the sums by the row
    epilogue (or by the fused SwiGLU epilogue).  Oracle: float64 on the DEQUANTISED weights, the bf16 pipeline's bar."""
    g = torch.Generator().manual_seed(M + N + K)
    w = torch.randn(N, K, generator=g) / math.sqrt(K) * (1.0 + torch.rand(N, 1, generator=g) * 3.0)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K, generator=g)
    rm, wq = ops.pack_rowmajor_fp8(w, bias, DEV)
    tl = ops.tiles16_from_rowmajor(rm)
    assert tl.wdtype == 2 and np.array_equal(tl.w.cpu().numpy(), ops.pack_tiles8_host(rm.w.cpu()))
    v = x.double() @ wq.double().T + bias.double()
    exp = F.silu(v[:, 0::2]) * v[:, 1::2] if glu else v
    R = ops.rows_R(M)
    planes = ops.rows_planes(R, K, DEV)
    ops.rows_finish(x.to(DEV), M, K, planes=planes, R=R)
    if fused:
        po = ops.rows_planes(R, N // 2, DEV)
        ops.rows_gemm(planes, tl, None, M, R, kgroups=1, glu_planes_out=po, glu_bias=rm.bias)
        torch.cuda.synchronize()
        got = _decode_planes(po, R, N // 2, False)[:M]
    else:
        kg = ops.rows_kgroups(N, K)
        part = torch.empty(kg, M, N, device=DEV)
        ops.rows_gemm(planes, tl, part, M, R, kgroups=kg)
        y = torch.empty(M, N // 2 if glu else N, device=DEV)
        ops.rows_finish(part, M, N, kg, bias=rm.bias, glu=glu, wscale=rm.scale, y=y)
        torch.cuda.synchronize()
        got = y.cpu()
    assert rel_err(got, exp) < 3e-5, rel_err(got, exp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,T,H,dh,pad", [(3, 77, 4, 64, 0), (2, 1500, 12, 64, 0), (2, 33, 2, 128, 32)])
def test_kv_head_major16_equals_the_permute_and_convert_passes(ops, dt, B, T, H, dh, pad):
    """mi355_kv_head_major16: the k | v projection rows [B, T, 2 H dh] -> two 16-bit head-major blocks [2, B, H, T, dh] in one pass, element for element what the reference's
    reshape / transpose / astype leave (whisper.py:360-365)."""
    g = torch.Generator().manual_seed(T + H)
    C = 2 * H * dh
    kv = torch.randn((B, T, C + pad), generator=g).to(DEV)[:, :, :C]
    got = ops.kv_head_major16(kv, 2, H, dh, dt)
    torch.cuda.synchronize()
    exp = kv.reshape(B, T, 2, H, dh).permute(2, 0, 3, 1, 4).to(dt)
    assert got.shape == (2, B, H, T, dh) and got.is_contiguous()
    assert torch.equal(got, exp)
