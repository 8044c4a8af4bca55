"""The deterministic cases of the reference's ``tests/vendor_parity`` suite, restated on the oracle and on the host mirrors.

The reference's suite is differential (vendored ``mlx_audio.lm`` vs upstream ``mlx_lm`` under one RNG seed, ``test_parity_sampling.py:19-29``);
neither MLX package exists in this image, so the same parameter grids are checked here against the DEFINITIONS the two implementations share
(``lm/sample_utils.py:130-234``, ``lm/models/cache.py:104-176, 502-717``), computed independently in float64 / with Python loops:

* ``apply_top_k`` k in {1, 5, 64, V-1} x batch {1, 3} x called twice  (test_parity_sampling.py:32-40)
* ``apply_top_p`` p in {0.1, 0.5, 0.9, 1.0} x batch {1, 3} x called twice  (:43-51)
* ``apply_min_p`` p in {0.05, 0.5}; ``min_p = 0`` raises ValueError from ``math.log(0)``  (:54-65)
* sampler cases (temp 0 / 1 / 0.7 with top-p, min-p, top-k, all three) over 20 steps with one noise tensor  (:78-101)
* ``KVCache.update_and_fetch`` step sequences [1]*8, [7,1,1,1], [256,1,1], [255,2,1], [512]  (test_parity_cache.py:36-47)
* ``BatchKVCache([0, 3])`` merge of extracted rows, wrapped and not; ``KVCache.merge``  (:83-112)
"""
import math

import numpy as np
import pytest
import torch

from mlx_audio_amd.lm.cache import BatchKVCache
from mlx_audio_amd.lm.stack import KVCache
from oracle import sampling_ref as S
from oracle.lm_ref import KVCacheRef

VOCAB = 128


def logprobs(batch=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, VOCAB, generator=g)
    return x - torch.logsumexp(x, dim=-1, keepdim=True)


@pytest.mark.parametrize("k", [1, 5, 64, VOCAB - 1])
@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("call_twice", [False, True])
def test_apply_top_k_cases(k, batch, call_twice):
    x = logprobs(batch)
    y = S.apply_top_k(x, k)
    if call_twice:
        assert torch.equal(S.apply_top_k(x, k), y)   # pure function of its input
        assert torch.equal(S.apply_top_k(y, k), y)   # and idempotent
    for b in range(batch):
        keep = set(np.argsort(-x[b].double().numpy(), kind="stable")[:k].tolist())
        for v in range(VOCAB):
            assert (float(y[b, v]) == float(x[b, v])) if v in keep else (float(y[b, v]) == -math.inf)


@pytest.mark.parametrize("p", [0.1, 0.5, 0.9, 1.0])
@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("call_twice", [False, True])
def test_apply_top_p_cases(p, batch, call_twice):
    x = logprobs(batch)
    y = S.apply_top_p(x, p)
    if call_twice:
        assert torch.equal(S.apply_top_p(x, p), y)
    for b in range(batch):
        # sample_utils.py:201-234: ascending cumulative probability; a token survives iff its cumulative mass exceeds 1 - p
        pr = np.exp(x[b].numpy().astype(np.float32))
        order = np.argsort(x[b].numpy(), kind="stable")
        cum = np.cumsum(pr[order], dtype=np.float32)
        keep = set(order[cum > np.float32(1 - p)].tolist())
        got = set(torch.nonzero(torch.isfinite(y[b])).flatten().tolist())
        assert got == keep
        assert len(got) >= 1 and int(torch.argmax(x[b])) in got
        kept_mass = float(np.exp(x[b].double().numpy())[sorted(got)].sum())
        assert kept_mass >= p - 1e-5
        if p == 1.0:
            assert len(got) == VOCAB
        assert torch.equal(y[b][torch.isfinite(y[b])], x[b][torch.isfinite(y[b])])


@pytest.mark.parametrize("p", [0.05, 0.5])
def test_apply_min_p_cases(p):
    x = logprobs(2)
    y = S.apply_min_p(x, p)
    for b in range(2):
        top = float(x[b].max())
        for v in range(VOCAB):
            dropped = float(x[b, v]) < np.float32(top + math.log(p))
            assert float(y[b, v]) == (-math.inf if dropped else float(x[b, v]))


def test_apply_min_p_zero_raises_like_the_reference():
    with pytest.raises(ValueError):  # math.log(0.0): "kept as a deliberate non-divergence" (test_parity_sampling.py:58-65)
        S.apply_min_p(logprobs(2), 0.0)


SAMPLER_CASES = [dict(temperature=0.0), dict(temperature=1.0), dict(temperature=0.7, top_p=0.9), dict(temperature=0.7, min_p=0.05),
                 dict(temperature=0.7, top_k=10), dict(temperature=0.7, top_p=0.95, top_k=20, min_p=0.02)]


@pytest.mark.parametrize("kw", SAMPLER_CASES)
def test_sampler_token_streams(kw):
    """Each of the 20 draws equals an independent float64 evaluation of the same chain (temperature -> top-k -> top-p -> min-p -> arg-max of
    filtered + Gumbel noise) wherever that decision is not a float32 knife edge, and is always a token the filters kept."""
    kw = dict(dict(top_k=0, top_p=1.0, min_p=0.0, repetition_penalty=1.0), **kw)
    g = torch.Generator().manual_seed(99)
    u = torch.rand(1, VOCAB, generator=g).clamp_(1e-20, 1 - 1e-7)
    noise = -(-u.log()).log()
    for step in range(20):
        x = logprobs(seed=step)
        tok = int(S.sample(x, gumbel=noise, **kw)[0])
        z = x[0].double().numpy()
        if kw["temperature"] <= 0:
            assert tok == int(np.argmax(z))
            continue
        z = z / kw["temperature"]
        keep = np.ones(VOCAB, bool)
        if 0 < kw["top_k"] < VOCAB:
            keep &= np.isin(np.arange(VOCAB), np.argsort(-z, kind="stable")[:kw["top_k"]])
        zz = np.where(keep, z, -np.inf)
        lp = zz - np.log(np.exp(zz - zz.max()).sum()) - zz.max()
        if 0.0 < kw["top_p"] < 1.0:
            order = np.argsort(lp, kind="stable")
            cum = np.cumsum(np.exp(lp[order]))
            kp = np.zeros(VOCAB, bool)
            kp[order[cum > 1 - kw["top_p"]]] = True
            edge = np.abs(cum - (1 - kw["top_p"])) < 1e-5
            keep &= kp | np.isin(np.arange(VOCAB), order[edge])  # float32 knife edges may fall either way
            lp = np.where(kp, lp, -np.inf)
        if kw["min_p"] > 0.0:
            keep &= ~(lp < lp.max() + math.log(kw["min_p"]))
        assert keep[tok], (step, tok)
        score = np.where(keep, z + noise[0].double().numpy(), -np.inf)
        top2 = np.sort(score)[-2:]
        if top2[1] - top2[0] > 1e-4:
            assert tok == int(np.argmax(score))


# ---------------------------------------------------------------------------------------------------------------- caches
G, DH = 4, 8
W = G * DH


def kv(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, n, W, generator=g), torch.randn(B, n, W, generator=g)


def to_ref(t):  # [B, n, G * dh] -> the reference's [B, G, n, dh]
    return t.reshape(t.shape[0], t.shape[1], G, DH).transpose(1, 2)


@pytest.mark.parametrize("steps", [[1] * 8, [7, 1, 1, 1], [256, 1, 1], [255, 2, 1], [512]])
def test_kvcache_update_and_fetch_sequences(steps):
    c, r = KVCache(G, DH, "cpu"), KVCacheRef()
    hist_k, hist_v = [], []
    for i, n in enumerate(steps):
        k, v = kv(2, n, i)
        hist_k.append(k)
        hist_v.append(v)
        slot = c.reserve(2, n)
        slot[:, :, :W], slot[:, :, W:] = k, v
        rk, rv = r.update_and_fetch(to_ref(k), to_ref(v))
        want_k, want_v = torch.cat(hist_k, 1), torch.cat(hist_v, 1)
        assert torch.equal(rk, to_ref(want_k)) and torch.equal(rv, to_ref(want_v))       # the oracle returns exactly what was appended
        assert torch.equal(c.keys, want_k) and torch.equal(c.values, want_v)             # and so does the device-layout mirror
        assert c.offset == r.offset == want_k.shape[1]
        # state (cache.py:140-150): capacity is a whole number of 256-steps past the last growth point; both agree on it
        assert c.kv.shape[1] == r.keys.shape[2] and c.kv.shape[1] >= c.offset
    want_cap = {(1,) * 8: 256, (7, 1, 1, 1): 256, (256, 1, 1): 512, (512,): 512,
                (255, 2, 1): 255 + 256}  # cache.py:122-124: a non-multiple prefix is sliced before the new block is appended
    assert c.kv.shape[1] == want_cap[tuple(steps)]


@pytest.mark.parametrize("wrapped", [False, True])
def test_batch_kvcache_merge_of_extracted_rows(wrapped):
    """test_parity_cache.py:83-100 (the continuous-batching round trip): ``BatchKVCache([0, 3])`` -> extract each row -> merge."""
    steps = [4, 1, 1] if wrapped else [2]
    b = BatchKVCache([0, 3], G, DH)
    hist = []
    for i, n in enumerate(steps):
        k, v = kv(2, n, 50 + i)
        b.update_and_fetch(k, v)
        hist.append((k, v))
    rows = [b.extract(i) for i in range(2)]
    total = sum(steps)
    assert [r.offset for r in rows] == [total, total - 3] if total > 3 else True
    m = BatchKVCache.merge(rows)
    lens = [r.offset for r in rows]
    assert m.left_padding.tolist() == [max(lens) - l for l in lens]
    assert m.offset.tolist() == lens and m.size() == max(lens)
    for i, r in enumerate(rows):
        p = int(m.left_padding[i])
        assert torch.equal(m.kv[i, p:p + r.offset], r.kv[0, :r.offset])
        assert float(m.kv[i, :p].abs().max()) == 0.0 if p else True
    allk = torch.cat([h[0] for h in hist], 1)
    assert torch.equal(rows[0].keys[0], allk[0])           # row 0 had no padding: its extract is its whole history
    assert torch.equal(rows[1].keys[0], allk[1, 3:])       # row 1: the 3 padded positions are dropped (cache.py:667-676)


def test_kvcache_merge_two_singles():
    singles = []
    for _ in range(2):
        c = KVCache(G, DH, "cpu")
        k, v = kv(1, 3, 1)
        s = c.reserve(1, 3)
        s[:, :, :W], s[:, :, W:] = k, v
        singles.append(c)
    m = BatchKVCache.merge(singles)
    assert m.left_padding.tolist() == [0, 0] and m.offset.tolist() == [3, 3]
    assert torch.equal(m.keys[0], singles[0].keys[0]) and torch.equal(m.keys[1], singles[1].keys[0])


def test_trim_roundtrip():
    c, r = KVCache(G, DH, "cpu"), KVCacheRef()
    for i, n in enumerate([5, 1]):
        k, v = kv(2, n, i)
        s = c.reserve(2, n)
        s[:, :, :W], s[:, :, W:] = k, v
        r.update_and_fetch(to_ref(k), to_ref(v))
    assert c.trim(2) == r.trim(2) == 2 and c.size() == r.offset == 4
    assert c.is_trimmable() and c.nbytes == c.kv.numel() * 4
