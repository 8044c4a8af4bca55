"""The sharded step over RCCL (backend "nccl", device tensors): world_size 2 on one node.  Needs two GPUs -- self-skips on the 1-GPU boxes the
round's ``pytest -m gpu`` runs on; the protocol itself is covered at world 2 / 3 / 8 over gloo in tests/test_shard_cpu.py.  A one-rank RCCL group
(runs on every box) sends the same collectives -- broadcast, int32 all_reduce, exact-size all_to_all_single on device tensors -- through the
"nccl" backend, so the calls themselves (dtypes, split lists, device placement) are exercised on RCCL even where no second GPU exists."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from mlx_audio_amd import shard
    from test_shard_cpu import SPF, FakeEngine, _make_requests, _ref_s_of, _single_process

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        reqs = _make_requests(9, 5) if rank == 0 else None
        ch = shard.ShardChannel(dev, dist, max_items=32, max_tokens=64)
        eng = FakeEngine(skew=True, device=dev)
        out = None
        for _ in range(2):
            out = shard.kokoro_step(ch, eng, reqs, lambda i, t: _ref_s_of(i, t).to(dev), SPF)
        torch.cuda.synchronize()
        if rank == 0:
            want = _single_process(reqs, True)
            # per step: broadcast + all_reduce + all_to_all back, + the re-balance all_to_all when another rank exists to take work
            ok = len(out) == 9 and all(o.is_cuda and torch.equal(o.cpu(), w) for o, w in zip(out, want))
            q.put((ok, ch.collectives))
    finally:
        dist.destroy_process_group()


def test_sharded_step_over_rccl_world2():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL over xGMI); the gloo tests cover the protocol")
    from test_shard_cpu import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get() == (True, 8)


def _worker_calls(port, q):
    """The exact collective calls ShardChannel issues (mlx_audio_amd/shard.py), on a one-rank RCCL group: request-block broadcast (int32), frame-count
    all_reduce (int32), exact-size all_to_all_single of float32 and float16 payloads with explicit split lists."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        block = torch.arange(4 * 64, dtype=torch.int32, device=dev).reshape(4, 64)
        dist.broadcast(block, src=0)
        counts = torch.tensor([3, 0, 7, 11], dtype=torch.int32, device=dev)
        dist.all_reduce(counts)
        ok = counts.tolist() == [3, 0, 7, 11] and int(block[3, 63]) == 255
        for dt in (torch.float32, torch.float16):
            src = torch.arange(1000, dtype=torch.float32, device=dev).to(dt)
            dst = torch.empty(1000, dtype=dt, device=dev)
            dist.all_to_all_single(dst, src, output_split_sizes=[1000], input_split_sizes=[1000])
            ok = ok and torch.equal(dst, src)
        # the autoregressive configs (shard.sharded_decode / scatter_dense): ragged int64 code sequences back, dense float32 requests out
        from mlx_audio_amd import shard

        ch = shard.ShardChannel(dev, None, max_items=8, max_tokens=16)
        ch.dist, ch.world, ch.rank = dist, 1, 0   # a one-rank group still issues every collective
        reqs = [torch.arange(3 + i, dtype=torch.int64) + i for i in range(4)]
        got = shard.sharded_decode(ch, reqs, lambda items, ids: [torch.arange(5 * (i + 1), dtype=torch.int64, device=dev) + 7 * i for i in items], dtype=torch.int64)
        ok = ok and len(got) == 4 and all(g.dtype == torch.int64 and torch.equal(g.cpu(), torch.arange(5 * (i + 1)) + 7 * i) for i, g in enumerate(got))
        x = torch.arange(3 * 10, dtype=torch.float32, device=dev).reshape(3, 10)
        mine, idx = shard.scatter_dense(ch, x, (10,))
        ok = ok and idx == [0, 1, 2] and torch.equal(mine, x) and ch.collectives == 4
        torch.cuda.synchronize()
        q.put(ok)
    finally:
        dist.destroy_process_group()


def test_collectives_go_through_rccl_on_a_one_rank_group():
    from test_shard_cpu import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_calls, args=(_free_port(), q))
    p.start()
    p.join(300)
    assert p.exitcode == 0 and q.get()
