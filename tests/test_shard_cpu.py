"""The multi-GPU path of the hot path is utterance sharding with one broadcast out and one gather back
(mlx_audio_amd/shard.py).  Exercised here with world_size 2 and 3 over ``gloo`` on CPU tensors; on the
GPU node the same code runs over RCCL (backend "nccl") with device tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_synth(ids: torch.Tensor) -> torch.Tensor:
    """Deterministic stand-in for the engine: 'waveform' length and content depend on the token ids."""
    n = 7 * int(ids.numel()) + int(ids.sum()) % 5
    return (torch.arange(n, dtype=torch.float32) * 0.25 + float(ids[0])) * (1.0 + float(ids.numel()))


def _make_requests(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(1, 178, (int(torch.randint(3, 40, (1,), generator=g)),), generator=g) for _ in range(n)]


def _worker(rank, world, port, n_utts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reqs = _make_requests(n_utts, 5) if rank == 0 else None
        ids, lens = shard.broadcast_requests(reqs, "cpu", dist)
        mine = shard.my_shard(lens, dist)
        audio = [_fake_synth(ids[i, : int(lens[i])].long()) for i in mine]
        out = shard.gather_waveforms(audio, mine, ids.shape[0], "cpu", dist)
        if rank == 0:
            ok = all(torch.equal(o, _fake_synth(r)) for o, r in zip(out, reqs)) and len(out) == n_utts
            q.put((ok, [len(shard.lpt_assign([int(v) for v in lens], world)[r]) for r in range(world)]))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_utts", [(2, 9), (3, 4), (2, 1)])
def test_broadcast_shard_gather_gloo(world, n_utts):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_utts, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, sizes = q.get()
    assert ok
    assert sum(sizes) == n_utts


def test_lpt_is_balanced_and_deterministic():
    costs = [264, 40, 300, 120, 90, 500, 33, 33, 260, 210, 75, 410]
    parts = shard.lpt_assign(costs, 4)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)  # LPT bound
    assert parts == shard.lpt_assign(costs, 4)
    assert shard.lpt_assign([], 2) == [[], []]
    assert shard.lpt_assign([5], 3) == [[0], [], []]


def test_single_process_passthrough():
    reqs = _make_requests(3, 1)
    ids, lens = shard.broadcast_requests(reqs, "cpu", None)
    assert ids.shape[0] == 3 and [int(v) for v in lens] == [r.numel() for r in reqs]
    mine = shard.my_shard(lens, None)
    assert mine == [0, 1, 2]
    out = shard.gather_waveforms([_fake_synth(r) for r in reqs], mine, 3, "cpu", None)
    assert all(torch.equal(o, _fake_synth(r)) for o, r in zip(out, reqs))


# ------------------------------------------------------------------ Whisper windows / Qwen3 batch items (dense inputs, integer outputs)
def _fake_transcribe(window: torch.Tensor) -> torch.Tensor:
    """Deterministic stand-in for the decoder: a token sequence whose length and content depend on the window."""
    n = 3 + int(window.abs().sum() * 10) % 7
    return (torch.arange(n, dtype=torch.int64) * 3 + int(window[0] * 100) % 50)


def _worker_dense(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        windows = torch.randn(n_items, 160, generator=g) if rank == 0 else None
        w = shard.broadcast_tensor(windows, "cpu", dist)
        mine = shard.lpt_assign([1] * w.shape[0], world)[rank]          # equal-cost items (30 s windows)
        toks = [_fake_transcribe(w[i]) for i in mine]
        out = shard.gather_waveforms(toks, mine, w.shape[0], "cpu", dist, dtype=torch.int64)
        if rank == 0:
            q.put(all(torch.equal(o, _fake_transcribe(windows[i])) and o.dtype == torch.int64 for i, o in enumerate(out)) and len(out) == n_items)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (3, 3)])
def test_dense_inputs_integer_outputs_gloo(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dense, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get()
