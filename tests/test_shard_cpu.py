"""The multi-GPU path of the hot path: utterance sharding through ``mlx_audio_amd.shard.ShardChannel`` -- one broadcast out, one tiny
all_reduce of the frame counts, an all_to_all re-balance when the real frame counts disagree with the token-count plan, one exact-size
all_to_all back.  Exercised here with world sizes 2, 3 and 8 over ``gloo`` on CPU tensors with a stand-in engine that has the Kokoro engine's
``front`` / ``back`` contract; on the GPU node the same code runs over RCCL (backend "nccl") with device tensors
(tests/test_shard_nccl_gpu.py needs >= 2 GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_amd import shard
from mlx_audio_amd.tts.models.kokoro.engine import KokoroFront

SPF = 6  # "samples per frame" of the stand-in vocoder


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    """Deterministic stand-in with the engine's split: ``front`` -> per-utterance state + data-dependent frame counts, ``back`` -> 'waveforms'
    of ``frames * SPF`` samples that depend on every field that crosses the split (so a wrong or corrupted move is visible)."""
    hid, sty = 5, 2

    def __init__(self, skew=False, device="cpu"):
        self.skew = skew
        self.device = torch.device(device)
        self.front_calls = 0
        self.back_items = []

    def frames_of(self, ids):
        base = 3 * int(ids.numel())
        if self.skew and int(ids[0]) % 4 == 0:
            base *= 9  # a few utterances are much longer in frames than their token count suggests
        return base + int(ids.sum()) % 4

    def front(self, ids, ref_s, forced_durations=None, speed=1.0):
        self.front_calls += 1
        dev = self.device
        ids = [i.to(dev) for i in ids]
        d = [(i.to(torch.float32)[:, None] * torch.arange(1, self.hid + self.sty + 1, dtype=torch.float32, device=dev)[None, :]) * 0.5 for i in ids]
        dur = [(i % 7 + 1).to(torch.int32) for i in ids]
        return KokoroFront([i.to(torch.int32) for i in ids], ref_s.to(dev), d, dur, [self.frames_of(i.cpu()) for i in ids], speed, None)

    def back(self, st):
        outs = []
        for b in range(len(st.ids)):
            n = st.frames[b] * SPF
            key = float(st.d[b].sum()) * 1e-3 + float(st.dur[b].sum()) + float(st.ref_s[b].sum()) + float(st.ids[b][0])
            outs.append(torch.arange(n, dtype=torch.float32, device=self.device) * 0.25 + key)
        self.back_items.append(len(st.ids))
        return outs, st.dur


def _ref_s_of(i, n_tokens):
    return torch.full((1, 2 * FakeEngine.sty), float(i) + 0.5 * n_tokens)


def _single_process(reqs, skew):
    eng = FakeEngine(skew)
    st = eng.front(reqs, torch.cat([_ref_s_of(i, int(r.numel())) for i, r in enumerate(reqs)], 0))
    return eng.back(st)[0]


def _make_requests(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(1, 178, (int(torch.randint(3, 40, (1,), generator=g)),), generator=g) for _ in range(n)]


def _worker(rank, world, port, n_utts, skew, wire, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reqs = _make_requests(n_utts, 5) if rank == 0 else None
        ch = shard.ShardChannel("cpu", dist, max_items=32, max_tokens=64)
        eng = FakeEngine(skew)
        token_plan = None
        for step in range(2):  # the channel is reused step after step
            out = shard.kokoro_step(ch, eng, reqs, _ref_s_of, SPF, tolerance=0.05, wire_dtype=wire)
            if step == 0:
                token_plan = shard.lpt_assign([int(r.numel()) for r in reqs], world) if rank == 0 else None
        if rank == 0:
            want = _single_process(reqs, skew)
            if wire is None:
                ok = len(out) == n_utts and all(torch.equal(o, w) for o, w in zip(out, want))
            else:
                ok = len(out) == n_utts and all(o.dtype == torch.float32 and torch.allclose(o, w, rtol=2e-3, atol=1e-2) for o, w in zip(out, want))
            q.put(dict(ok=ok, collectives=ch.collectives, plan=ch.owned, token_plan=token_plan,
                       frames=[eng.frames_of(r) for r in reqs]))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _run(world, n_utts, skew=False, wire=None):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_utts, skew, wire, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return q.get()


@pytest.mark.parametrize("world,n_utts", [(2, 9), (3, 4), (2, 1), (8, 19), (8, 5)])
def test_sharded_step_equals_single_process(world, n_utts):
    r = _run(world, n_utts)
    assert r["ok"]
    assert sorted(i for o in r["plan"] for i in o) == list(range(n_utts))
    # per step: broadcast + all_reduce(frame counts) + all_reduce(failure flag) + all_to_all(waveforms); frames ~ 3 x tokens here, so nothing needs to move
    assert r["plan"] == r["token_plan"]
    assert r["collectives"] == 2 * 4


@pytest.mark.parametrize("world,n_utts", [(2, 9), (3, 11), (8, 19)])
def test_rebalance_on_real_frame_counts(world, n_utts):
    r = _run(world, n_utts, skew=True)
    assert r["ok"]
    frames = r["frames"]
    assert r["plan"] != r["token_plan"]                                   # utterances moved ...
    assert shard.makespan(frames, r["plan"]) < shard.makespan(frames, r["token_plan"])   # ... and the slowest rank got faster
    assert shard.makespan(frames, r["plan"]) <= 1.05 * shard.makespan(frames, shard.lpt_assign(frames, world)) + max(frames) * 0
    assert r["collectives"] == 2 * 5                                      # one extra all_to_all per step, nothing else


def test_fp16_on_the_wire():
    assert _run(2, 6, wire=torch.float16)["ok"]


def test_lpt_is_balanced_and_deterministic():
    costs = [264, 40, 300, 120, 90, 500, 33, 33, 260, 210, 75, 410]
    parts = shard.lpt_assign(costs, 4)
    loads = [sum(costs[i] for i in p) for p in parts]
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    assert max(loads) - min(loads) <= max(costs) // 2
    assert parts == shard.lpt_assign(costs, 4)
    assert shard.lpt_assign([], 2) == [[], []]
    assert shard.lpt_assign([5], 3) == [[0], [], []]


def test_lpt_rebalance_moves_only_when_it_pays():
    costs = [100, 100, 100, 100]
    cur = [[0, 1], [2, 3]]
    assert shard.lpt_rebalance(costs, cur) == cur                        # already optimal: nothing moves, whatever fresh LPT would pick
    cur = [[0, 2], [1, 3]]
    assert shard.lpt_rebalance(costs, cur) == cur
    costs = [900, 100, 100, 100]
    new = shard.lpt_rebalance(costs, [[0, 1], [2, 3]])
    assert new == [[0], [1, 2, 3]]                                        # item 1 moves, items 2 / 3 stay where they were
    assert shard.lpt_rebalance([10, 11, 10, 10], [[0, 1], [2, 3]], tolerance=0.10) == [[0, 1], [2, 3]]


def test_single_process_path_and_capacity_errors():
    reqs = _make_requests(3, 9)
    ch = shard.ShardChannel("cpu", None, max_items=4, max_tokens=64)
    out = shard.kokoro_step(ch, FakeEngine(), reqs, _ref_s_of, SPF)
    want = _single_process(reqs, False)
    assert ch.collectives == 0 and all(torch.equal(o, w) for o, w in zip(out, want))
    with pytest.raises(ValueError):
        ch.scatter_requests(_make_requests(5, 1))
    with pytest.raises(ValueError):
        shard.ShardChannel("cpu", None, max_items=4, max_tokens=2).scatter_requests(reqs)


def test_front_state_pack_roundtrip():
    eng = FakeEngine()
    reqs = _make_requests(4, 3)
    st = eng.front(reqs, torch.cat([_ref_s_of(i, int(r.numel())) for i, r in enumerate(reqs)], 0))
    blobs = [st.pack(i) for i in range(4)]
    width, style = eng.hid + eng.sty, 2 * eng.sty
    assert [int(b.numel()) for b in blobs] == [KokoroFront.packed_size(int(r.numel()), style, width) for r in reqs]
    back = KokoroFront.unpack(blobs, st.frames, style, width)
    for i in range(4):
        assert torch.equal(back.ids[i], st.ids[i]) and torch.equal(back.dur[i], st.dur[i]) and torch.equal(back.d[i], st.d[i])
    assert torch.equal(back.ref_s, st.ref_s)
    sel = st.select([2, 0])
    assert torch.equal(sel.ids[0], st.ids[2]) and sel.frames == [st.frames[2], st.frames[0]] and torch.equal(sel.ref_s[1], st.ref_s[0])


def _worker_windows(rank, world, port, q):
    """The Whisper-shaped use: equal-cost dense items (30 s windows) out with a known shape (one broadcast), ragged token lists back."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        windows = torch.randn(5, 8, 4, generator=g) if rank == 0 else None
        w = shard.broadcast_tensor(windows, "cpu", dist, shape=(5, 8, 4))
        ch = shard.ShardChannel("cpu", dist, max_items=8, max_tokens=1)
        ch.n_items, ch.owned = 5, shard.lpt_assign([1] * 5, world)
        toks = [torch.arange(3 + i, dtype=torch.int64) * (i + 1) + int(w[i].abs().sum() * 10) for i in ch.my_items()]
        out = ch.gather(toks, dtype=torch.int64)
        if rank == 0:
            want = [torch.arange(3 + i, dtype=torch.int64) * (i + 1) + int(windows[i].abs().sum() * 10) for i in range(5)]
            q.put(all(torch.equal(a, b) for a, b in zip(out, want)) and ch.collectives == 2)
    finally:
        dist.destroy_process_group()


def test_dense_windows_out_ragged_tokens_back():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_windows, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get()


# ------------------------------------------------------------------------------------------------ the autoregressive configs: ragged integer results
def _decode_worker(rank, world, port, n_seq, gather, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ch = shard.ShardChannel("cpu", dist, max_items=64, max_tokens=40)
        g = torch.Generator().manual_seed(3)
        reqs = [torch.randint(1, 400, (int(torch.randint(3, 30, (1,), generator=g)),), generator=g) for _ in range(n_seq)]

        def run_local(items, ids):   # "code frames" of a sequence: ragged [frames, 4] int64 that depend on its ids and on its GLOBAL index
            return [(torch.arange((int(t.sum()) % 7 + 1) * 4, dtype=torch.int64) + 1000 * i + int(t[0])) for i, t in zip(items, ids)]

        want = run_local(list(range(n_seq)), [t.to(torch.int32) for t in reqs])
        out = shard.sharded_decode(ch, reqs if rank == 0 else None, run_local, dtype=torch.int64, gather=gather)
        if gather == "none":
            mine, local = out
            ok = mine == ch.my_items() and all(torch.equal(a, want[i]) for i, a in zip(mine, local))
            got = dict(rank=rank, mine=mine, ok=bool(ok), collectives=ch.collectives)
            allg = [None] * world
            dist.all_gather_object(allg, got)
            if rank == 0:
                q.put(dict(ok=all(x["ok"] for x in allg), items=sorted(i for x in allg for i in x["mine"]), collectives=allg[0]["collectives"]))
        else:
            if rank == 0:
                ok = len(out) == n_seq and all(o.dtype == torch.int64 and torch.equal(o, w) for o, w in zip(out, want))
                q.put(dict(ok=bool(ok), items=list(range(n_seq)), collectives=ch.collectives))
            else:
                assert out is None
        # dense requests (Whisper windows): one broadcast, round-robin plan, every rank sees exactly its rows
        x = torch.arange(n_seq * 6, dtype=torch.float32).reshape(n_seq, 2, 3) if rank == 0 else None
        mine_x, idx = shard.scatter_dense(ch, x, (2, 3))
        full = torch.arange(n_seq * 6, dtype=torch.float32).reshape(n_seq, 2, 3)
        assert idx == ch.my_items() and torch.equal(mine_x, full[idx])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_seq,gather", [(2, 9, "rank0"), (8, 19, "rank0"), (8, 5, "rank0"), (2, 9, "none"), (8, 19, "none")])
def test_sharded_decode_ragged_integer_results(world, n_seq, gather):
    """``shard.sharded_decode``: code / token sequences of different lengths come back bit-exact and in request order (rank0), or stay on the rank
    that made them (none: no result collective at all)."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_decode_worker, args=(r, world, port, n_seq, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    r = q.get()
    assert r["ok"] and r["items"] == list(range(n_seq))
    # broadcast + (rank0: all_reduce of the result lengths + all_to_all of the payload)
    assert r["collectives"] == (3 if gather == "rank0" else 1)


# ---- a step that fails on one rank: every rank leaves it together and the group stays usable (round-3 advisor finding: a rank-0 exception between
# two collectives used to strand the workers inside the step)
class _FlakyEngine(FakeEngine):
    def __init__(self, fail_front_on=None, fail_back_on=None):
        super().__init__(False)
        self.calls = 0
        self.fail_front_on, self.fail_back_on = fail_front_on, fail_back_on

    def front(self, *a, **k):
        self.calls += 1
        if self.fail_front_on == self.calls:
            raise RuntimeError("front exploded")
        return super().front(*a, **k)

    def back(self, *a, **k):
        if self.fail_back_on == self.calls:
            raise RuntimeError("back exploded")
        return super().back(*a, **k)


def _failing_worker(rank, world, port, bad_rank, where, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reqs = _make_requests(7, 5) if rank == 0 else None
        ch = shard.ShardChannel("cpu", dist, max_items=32, max_tokens=64)
        eng = _FlakyEngine(**({("fail_front_on" if where == "front" else "fail_back_on"): 2} if rank == bad_rank else {}))
        results = []
        for step in range(3):   # step 2 (the engine's second call) fails on bad_rank; steps 1 and 3 must be served normally
            try:
                out = shard.kokoro_step(ch, eng, reqs, _ref_s_of, SPF)
                results.append("ok" if (rank != 0 or all(torch.equal(o, w) for o, w in zip(out, _single_process(reqs, False)))) else "wrong")
            except shard.ShardStepFailed as e:
                results.append("failed-local" if e.__cause__ is not None else "failed-remote")
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank,where", [(0, "front"), (1, "front"), (1, "back"), (0, "back")])
def test_failed_step_is_left_by_every_rank_and_the_group_survives(bad_rank, where):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, bad_rank, where, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0   # nobody hangs in a collective, nobody dies
    got = dict(q.get() for _ in range(world))
    for r in range(world):
        assert got[r] == ["ok", "failed-local" if r == bad_rank else "failed-remote", "ok"], got


def test_sharded_forward_validates_before_the_header():
    """Everything checkable is refused BEFORE the CMD_RUN header: no collective has been issued when the exception leaves forward()."""
    ch = shard.ShardChannel("cpu", None, max_items=4, max_tokens=16)
    sk = shard.ShardedKokoro(FakeEngine(False), ch, SPF)
    ok_ids = [torch.randint(1, 100, (5,)) for _ in range(2)]
    style = torch.zeros(2, 2 * FakeEngine.sty)
    for ids, ref, speed in (([torch.randint(1, 100, (5,)) for _ in range(5)], torch.zeros(5, 2 * FakeEngine.sty), 1.0),   # too many requests
                            ([torch.randint(1, 100, (17,))], torch.zeros(1, 2 * FakeEngine.sty), 1.0),                      # too long
                            (ok_ids, torch.zeros(3, 2 * FakeEngine.sty), 1.0),                                              # style rows != requests
                            (ok_ids, style, 0.0), ([], torch.zeros(0, 2 * FakeEngine.sty), 1.0)):
        before = ch.collectives
        with pytest.raises(ValueError):
            sk.forward(ids, ref, speed=speed)
        assert ch.collectives == before and sk.steps == 0
    outs, _ = sk.forward(ok_ids, style)
    assert len(outs) == 2 and sk.steps == 1
