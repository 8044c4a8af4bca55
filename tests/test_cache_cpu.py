"""KVCache / BatchKVCache: the device-layout mirrors against the restated reference classes
(oracle/lm_ref.py KVCacheRef = lm/models/cache.py:104-176) and against the invariants the reference's continuous batching relies on
(cache.py:606-717: merge -> step -> extract round trips, filter, extend).  Pure tensor plumbing: runs on CPU."""
import torch

from mlx_audio_amd.lm.cache import BatchKVCache
from mlx_audio_amd.lm.stack import KVCache
from oracle.lm_ref import KVCacheRef

G, DH = 2, 4
W = G * DH


def _kv(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, n, W, generator=g), torch.randn(B, n, W, generator=g)


def _to_ref(t):  # [B, n, G*dh] -> [B, G, n, dh] (the reference's layout)
    return t.reshape(t.shape[0], t.shape[1], G, DH).transpose(1, 2)


def test_kvcache_matches_reference_growth_and_views():
    c, r = KVCache(G, DH, "cpu"), KVCacheRef()
    for i, n in enumerate((3, 1, 1, 260, 1, 300)):
        k, v = _kv(2, n, i)
        slot = c.reserve(2, n)
        slot[:, :, :W] = k
        slot[:, :, W:] = v
        rk, rv = r.update_and_fetch(_to_ref(k), _to_ref(v))
        assert c.offset == r.offset
        assert c.kv.shape[1] == r.keys.shape[2]          # same step-256 capacity schedule
        assert torch.equal(_to_ref(c.keys), rk) and torch.equal(_to_ref(c.values), rv)
    assert c.trim(5) == r.trim(5) and c.offset == r.offset
    assert c.trim(10 ** 6) == r.trim(10 ** 6) == 561 and c.offset == 0


def test_batch_cache_merge_step_extract_roundtrip():
    singles = []
    for i, n in enumerate((5, 2, 9)):
        c = KVCache(G, DH, "cpu")
        k, v = _kv(1, n, 10 + i)
        s = c.reserve(1, n)
        s[:, :, :W], s[:, :, W:] = k, v
        singles.append(c)
    b = BatchKVCache.merge(singles)
    assert b.left_padding.tolist() == [4, 7, 0] and b.offset.tolist() == [5, 2, 9] and b.size() == 9
    k, v = _kv(3, 1, 99)
    keys, vals = b.update_and_fetch(k, v)           # one decode step for all three requests
    assert keys.shape == (3, 10, W) and b.offset.tolist() == [6, 3, 10]
    for i, c in enumerate(singles):
        e = b.extract(i)
        assert e.offset == c.offset + 1
        assert torch.equal(e.kv[0, :c.offset], c.kv[0, :c.offset])         # history preserved
        assert torch.equal(e.kv[0, c.offset, :W], k[i, 0]) and torch.equal(e.kv[0, c.offset, W:], v[i, 0])
    assert float(b.kv[1, :7].abs().max()) == 0.0     # left padding stays zero: the kernel's k_start skips it


def test_batch_cache_filter_and_extend():
    b = BatchKVCache([1, 3, 0], G, DH)
    k, v = _kv(3, 4, 1)
    b.update_and_fetch(k, v)
    assert b.offset.tolist() == [3, 1, 4] and b._idx == 4
    b.filter([0, 1])                                  # drop the unpadded row: everything shifts left by min padding
    assert b.left_padding.tolist() == [0, 2] and b._idx == 3 and b.kv.shape[0] == 2
    assert torch.equal(b.keys[0], k[0, 1:]) and torch.equal(b.keys[1, 2:], k[1, 3:])
    other = BatchKVCache([0], G, DH)
    k2, v2 = _kv(1, 6, 2)
    other.update_and_fetch(k2, v2)
    b.extend(other)                                   # right-justify both to the longer one
    assert b._idx == 6 and b.left_padding.tolist() == [3, 5, 0] and b.offset.tolist() == [3, 1, 6]
    assert torch.equal(b.keys[2], k2[0]) and torch.equal(b.keys[0, 3:], k[0, 1:])
    assert b.trim(2) == 2 and b._idx == 4 and b.offset.tolist() == [1, -1, 4]
    e1, e2 = BatchKVCache([2], G, DH), BatchKVCache([0, 1], G, DH)
    e1.extend(e2)
    assert e1.left_padding.tolist() == [2, 0, 1] and e1.empty()


def _chk(t):
    t = t.double()
    return [list(t.shape), float(t.sum()), float((t ** 2).sum())]


def _same(a, b, tol=1e-9):
    return a[0] == b[0] and abs(a[1] - b[1]) <= tol * (1 + abs(b[1])) and abs(a[2] - b[2]) <= tol * (1 + b[2])


def test_caches_against_the_reference_classes_own_run():
    """tests/golden/ref_cache.json = the reference's OWN ``KVCache`` / ``BatchKVCache`` (lm/models/cache.py, executed over the numpy stand-in for MLX by
    tests/golden/make_reference_fixtures.py) through the operation sequences of this file: this package's device-layout mirrors reproduce every offset,
    capacity, left padding, trim count and the fetched contents."""
    import json
    import os

    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cache.json")))
    c = KVCache(G, DH, "cpu")
    for i, (n, row) in enumerate(zip((3, 1, 1, 260, 1, 300), want["kvcache"]["rows"])):
        k, v = _kv(2, n, i)
        slot = c.reserve(2, n)
        slot[:, :, :W], slot[:, :, W:] = k, v
        assert c.offset == row["offset"] and c.kv.shape[1] == row["capacity"]
        assert _same(_chk(_to_ref(c.keys)), row["keys"]) and _same(_chk(_to_ref(c.values)), row["values"])
    kw = want["kvcache"]
    assert c.trim(5) == kw["trim5"] and c.offset == kw["offset_after"] and c.trim(10 ** 6) == kw["trim_all"] and c.offset == kw["offset_end"]

    singles = []
    for i, n in enumerate((5, 2, 9)):
        s = KVCache(G, DH, "cpu")
        k, v = _kv(1, n, 10 + i)
        sl = s.reserve(1, n)
        sl[:, :, :W], sl[:, :, W:] = k, v
        singles.append(s)
    b = BatchKVCache.merge(singles)
    m = want["merge"]
    assert b.left_padding.tolist() == m["left_padding"] and b.offset.tolist() == m["offset"] and b.size() == m["size"]
    k, v = _kv(3, 1, 99)
    keys, _ = b.update_and_fetch(k, v)
    assert _same(_chk(_to_ref(keys)), m["step_keys"]) and b.offset.tolist() == m["step_offset"]
    for i, e in enumerate(m["extracted"]):
        x = b.extract(i)
        assert x.offset == e["offset"] and _same(_chk(_to_ref(x.keys)), e["keys"])

    f = want["filter_extend"]
    b = BatchKVCache([1, 3, 0], G, DH)
    k, v = _kv(3, 4, 1)
    b.update_and_fetch(k, v)
    assert b.offset.tolist() == f["offset0"] and b._idx == f["idx0"]
    b.filter([0, 1])
    assert b.left_padding.tolist() == f["left_padding1"] and b._idx == f["idx1"] and _same(_chk(_to_ref(b.keys)), f["keys1"])
    other = BatchKVCache([0], G, DH)
    k2, v2 = _kv(1, 6, 2)
    other.update_and_fetch(k2, v2)
    b.extend(other)
    assert b._idx == f["idx2"] and b.left_padding.tolist() == f["left_padding2"] and b.offset.tolist() == f["offset2"] and _same(_chk(_to_ref(b.keys)), f["keys2"])
    assert b.trim(2) == f["trim"] and b._idx == f["idx3"] and b.offset.tolist() == f["offset3"]
    e1, e2 = BatchKVCache([2], G, DH), BatchKVCache([0, 1], G, DH)
    e1.extend(e2)
    assert e1.left_padding.tolist() == f["empty_extend_left_padding"] and e1.empty() == f["empty"]
