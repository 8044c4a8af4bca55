"""Utterance / window sharding over the GPUs of one node (one process per GPU, ``torch.distributed``).

The hot path shards naturally: utterances are independent (Kokoro chunks of
<= 510 phonemes, ``tts/models/kokoro/pipeline.py:266-293``), weights are replicated, and the only
exchanges are tiny: the padded int32 token batch goes out from rank 0 (broadcast), waveforms come
back (gather).  The reference has no counterpart (single device).  No tensor / sequence
parallelism: the largest model of the path is 3.4 GB.

Everything here is backend-agnostic: the same code runs over RCCL (backend ``"nccl"`` on ROCm, GPU
tensors, xGMI) and over ``gloo`` (CPU tensors; the world_size-2 tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def lpt_assign(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of work items to ``world`` ranks.

    Returns, per rank, the (ascending) indices of the items it owns.  Deterministic (ties broken by
    index / lowest rank) so that every rank computes the same partition without communication."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += int(costs[i])
    return [sorted(o) for o in owned]


def pack_token_batch(ids_list: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """List of 1-D integer tensors -> (padded int32 ``[N, Tmax]``, int32 lengths ``[N]``)."""
    n = len(ids_list)
    lens = torch.tensor([int(t.numel()) for t in ids_list], dtype=torch.int32)
    tmax = int(lens.max()) if n else 0
    out = torch.zeros((n, tmax), dtype=torch.int32)
    for i, t in enumerate(ids_list):
        out[i, : t.numel()] = t.to(torch.int32)
    return out, lens


def broadcast_requests(ids_list: Optional[Sequence[torch.Tensor]], device, dist=None, src: int = 0):
    """Rank ``src`` owns the request batch; afterwards every rank holds (padded ids, lens) on ``device``."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        ids, lens = pack_token_batch(ids_list)
        return ids.to(device), lens.to(device)
    rank = dist.get_rank()
    hdr = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        ids, lens = pack_token_batch(ids_list)
        hdr[0], hdr[1] = ids.shape[0], ids.shape[1]
    dist.broadcast(hdr, src)
    n, tmax = int(hdr[0]), int(hdr[1])
    if rank == src:
        ids, lens = ids.to(device), lens.to(device)
    else:
        ids = torch.empty((n, tmax), dtype=torch.int32, device=device)
        lens = torch.empty((n,), dtype=torch.int32, device=device)
    dist.broadcast(ids, src)
    dist.broadcast(lens, src)
    return ids, lens


def my_shard(lens: torch.Tensor, dist=None, cost=None) -> List[int]:
    """Indices of the utterances this rank synthesises (LPT over a per-utterance cost; default: token count,
    a proxy for the frame count that is only known after the duration predictor ran)."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    costs = [int(v) for v in (cost if cost is not None else lens.cpu())]
    return lpt_assign(costs, world)[rank]


def broadcast_tensor(t: Optional[torch.Tensor], device, dist=None, src: int = 0, dtype=torch.float32) -> torch.Tensor:
    """Rank ``src`` owns a dense tensor (e.g. the 30 s audio windows of a Whisper request, the prefill embeddings of a Qwen3 batch);
    afterwards every rank holds it on ``device``.  Shape travels first (one int64 header), then the payload."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t.to(device=device, dtype=dtype)
    rank = dist.get_rank()
    hdr = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == src:
        assert t.dim() <= 7
        hdr[0] = t.dim()
        for i, n in enumerate(t.shape):
            hdr[1 + i] = n
    dist.broadcast(hdr, src)
    shape = [int(v) for v in hdr[1:1 + int(hdr[0])]]
    buf = t.to(device=device, dtype=dtype).contiguous() if rank == src else torch.empty(shape, dtype=dtype, device=device)
    dist.broadcast(buf, src)
    return buf


def gather_waveforms(local_audio: Sequence[torch.Tensor], local_idx: Sequence[int], n_total: int, device, dist=None,
                     dst: int = 0, dtype=torch.float32) -> Optional[List[torch.Tensor]]:
    """Collects every rank's per-item 1-D results (waveforms; with ``dtype=torch.int64`` token / code sequences) on rank ``dst`` in
    original item order.

    One all-gather of the per-utterance sample counts (int64 x n_total), then one padded gather of float32
    samples ``[n_local_max, samples_max]`` per rank.  Returns the list on ``dst`` and ``None`` elsewhere."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world == 1:
        out: List[Optional[torch.Tensor]] = [None] * n_total
        for i, a in zip(local_idx, local_audio):
            out[i] = a
        return out  # type: ignore[return-value]
    rank = dist.get_rank()
    counts = torch.zeros(n_total, dtype=torch.int64, device=device)
    for i, a in zip(local_idx, local_audio):
        counts[i] = a.numel()
    dist.all_reduce(counts)  # disjoint ownership: sum == all-gather of the per-utterance counts
    counts_h = counts.cpu()
    smax = int(counts_h.max()) if n_total else 0
    # every rank derives every rank's ownership from the counts it contributed: gather the index lists
    nloc = torch.tensor([len(local_idx)], dtype=torch.int64, device=device)
    nlocs = [torch.zeros_like(nloc) for _ in range(world)]
    dist.all_gather(nlocs, nloc)
    nmax = max(int(v) for v in nlocs)
    idx_pad = torch.full((nmax,), -1, dtype=torch.int64, device=device)
    if local_idx:
        idx_pad[: len(local_idx)] = torch.tensor(list(local_idx), dtype=torch.int64, device=device)
    payload = torch.zeros((nmax, smax), dtype=dtype, device=device)
    for j, a in enumerate(local_audio):
        payload[j, : a.numel()] = a.reshape(-1).to(device=device, dtype=dtype)
    if rank == dst:
        idx_all = [torch.empty_like(idx_pad) for _ in range(world)]
        pay_all = [torch.empty_like(payload) for _ in range(world)]
    else:
        idx_all = pay_all = None
    dist.gather(idx_pad, idx_all, dst=dst)
    dist.gather(payload, pay_all, dst=dst)
    if rank != dst:
        return None
    out = [None] * n_total
    for r in range(world):
        for j, i in enumerate(idx_all[r].cpu().tolist()):
            if i >= 0:
                out[i] = pay_all[r][j, : int(counts_h[i])]
    return out  # type: ignore[return-value]
