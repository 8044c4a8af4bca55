"""Utterance / window sharding over the GPUs of one node (one process per GPU, ``torch.distributed``).

The hot path shards naturally: utterances are independent (Kokoro chunks of <= 510 phonemes,
``tts/models/kokoro/pipeline.py:266-293``), weights are replicated, and the exchanges are small: token ids go out from the rank that
owns the request queue, waveforms come back.  The reference has no counterpart (single device).  No tensor / sequence parallelism:
the largest model of the path is 3.4 GB.

Wire protocol of one step (``ShardChannel``; every message is ONE collective, sizes never travel in a separate message):

1. ``scatter_requests``: ONE ``broadcast`` of a fixed-capacity int32 block ``[max_items, 1 + max_tokens]`` (column 0 = token count, -1 = empty
   row).  The capacity is agreed when the channel is created, so no header round trip.  The token counts are read back to the host once
   (the host needs the shapes to build its batch): the only host sync this module adds.
2. Every rank derives the same longest-processing-time assignment from the token counts (``lpt_assign``: deterministic, no communication).
3. After the model's token-rate half (Kokoro: PL-BERT + duration predictor) each rank knows the true frame counts of ITS utterances.
   ``share_counts``: ONE ``all_reduce`` of an int32 vector ``[max_items]`` (disjoint ownership: sum == all-gather) makes them global.  The model
   has just synced on those counts itself (they size its buffers), so reading the reduced vector costs a copy, not a pipeline bubble.
4. ``rebalance`` (optional): LPT again on the real frame counts; utterances whose owner changes move as packed float32 blobs in ONE
   ``all_to_all_single`` with exact split sizes (no padding).  Skipped when the token-count plan is already within ``tolerance`` of the
   frame-count plan's makespan.
5. ``gather``: ONE ``all_to_all_single`` towards the destination rank with exact split sizes (every rank can compute every rank's sample counts
   from step 3 -- samples = frames x 600 for Kokoro -- or passes ``counts``), optionally as fp16 / int16 on the wire.  No padding crosses xGMI.

Everything here is backend-agnostic: the same code runs over RCCL (backend ``"nccl"`` on ROCm, GPU tensors, xGMI) and over ``gloo`` (CPU
tensors; the world_size 2 / 3 / 8 tests).  xGMI note: a gather to one rank is bound by that rank's 7 inbound links
(~153 GB/s each); 64 utterances x 634 KB per rank is 40 MB per link per step, ~0.3 ms against a ~60 ms step.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

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


def makespan(costs: Sequence[int], owned: Sequence[Sequence[int]]) -> int:
    return max((sum(int(costs[i]) for i in o) for o in owned), default=0)


def lpt_rebalance(costs: Sequence[int], current: Sequence[Sequence[int]], tolerance: float = 0.05) -> List[List[int]]:
    """Plan on the real costs, moving as few items as possible: keep ``current`` when its makespan is within ``tolerance`` of a fresh LPT plan's;
    otherwise LPT with ties broken towards the current owner (so an item only moves when that shortens the schedule)."""
    world = len(current)
    fresh = lpt_assign(costs, world)
    if makespan(costs, current) <= (1.0 + tolerance) * makespan(costs, fresh):
        return [sorted(o) for o in current]
    owner = {i: r for r, o in enumerate(current) for i in o}
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], 0 if owner.get(i) == k else 1, k))
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


class ShardChannel:
    """The per-process-group state of the protocol above: capacities, the request block, the last plan."""

    def __init__(self, device, dist=None, max_items: int = 1024, max_tokens: int = 512, src: int = 0, dst: int = 0):
        self.device = torch.device(device)
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.max_items, self.max_tokens, self.src, self.dst = int(max_items), int(max_tokens), src, dst
        self.block = torch.full((self.max_items, 1 + self.max_tokens), -1, dtype=torch.int32, device=self.device)
        self.n_items = 0
        self.owned: List[List[int]] = [[] for _ in range(self.world)]
        self.collectives = 0  # issued by this channel since creation (the tests count them)

    # ------------------------------------------------------------------ requests out
    def scatter_requests(self, ids_list: Optional[Sequence[torch.Tensor]]) -> Tuple[torch.Tensor, List[int]]:
        """Rank ``src`` passes the request batch (others ``None``).  Returns (block ``[n, 1 + max_tokens]`` on the device: column 0 = length,
        then the ids; token counts as host ints) on every rank and stores the token-count LPT plan in ``self.owned``."""
        if self.rank == self.src:
            n = len(ids_list)
            if n > self.max_items or any(int(t.numel()) > self.max_tokens for t in ids_list):
                raise ValueError(f"request batch exceeds the channel capacity ({self.max_items} items x {self.max_tokens} tokens)")
            host = torch.full((self.max_items, 1 + self.max_tokens), -1, dtype=torch.int32)
            if n:
                pad = torch.nn.utils.rnn.pad_sequence([t.to(dtype=torch.int32, device="cpu") for t in ids_list], batch_first=True)
                host[:n, 0] = torch.tensor([int(t.numel()) for t in ids_list], dtype=torch.int32)
                host[:n, 1:1 + pad.shape[1]] = pad
            self.block.copy_(host, non_blocking=True)
        if self.dist:
            self.dist.broadcast(self.block, self.src)
            self.collectives += 1
        if self.dist:
            lens_all = self.block[:, 0].cpu()  # the one read-back: the host builds its batch from these shapes
            n = int((lens_all >= 0).sum())
            lens = [int(v) for v in lens_all[:n]]
        else:
            lens = [int(t.numel()) for t in ids_list]  # single process: the shapes never left the host
            n = len(lens)
        self.n_items = n
        self.owned = lpt_assign(lens, self.world)
        return self.block[:n], lens

    def my_items(self) -> List[int]:
        return self.owned[self.rank]

    def my_ids(self, block: torch.Tensor, lens: Sequence[int]) -> List[torch.Tensor]:
        return [block[i, 1:1 + lens[i]] for i in self.my_items()]

    # ------------------------------------------------------------------ counts
    def share_counts(self, local_counts: Sequence[int], owned: Optional[Sequence[Sequence[int]]] = None) -> List[int]:
        """Per-item integer (frame / sample / token count) known only to the item's owner -> the full vector on every rank.  ONE all_reduce."""
        owned = self.owned if owned is None else owned
        mine = owned[self.rank]
        assert len(local_counts) == len(mine), (len(local_counts), len(mine))
        vec = torch.zeros(self.max_items, dtype=torch.int32)
        for i, c in zip(mine, local_counts):
            vec[i] = int(c)
        if not self.dist:
            return [int(v) for v in vec[: self.n_items]]
        dev_vec = vec.to(self.device, non_blocking=True)
        self.dist.all_reduce(dev_vec)
        self.collectives += 1
        return [int(v) for v in dev_vec[: self.n_items].cpu()]

    def any_failed(self, failed: bool) -> bool:
        """Did the step fail on ANY rank?  One int32 all_reduce (a rank that raised keeps taking part in the protocol until every rank knows)."""
        if not self.dist:
            return bool(failed)
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(flag)
        self.collectives += 1
        return bool(int(flag.cpu()[0]) > 0)

    # ------------------------------------------------------------------ re-balance on the real costs
    def rebalance(self, costs: Sequence[int], pack: Callable[[int], torch.Tensor], blob_size: Callable[[int], int],
                  tolerance: float = 0.05) -> Tuple[List[List[int]], dict]:
        """``costs``: real per-item costs (all items, e.g. ``share_counts`` of the frame counts).  ``pack(i)`` -> 1-D float32 blob of item ``i`` (called
        only for items this rank gives away), ``blob_size(i)`` -> its length (computable by every rank).  Returns (new plan, {item: blob}
        for the items this rank received).  ONE all_to_all_single when anything moves, no collective otherwise."""
        new = lpt_rebalance(costs, self.owned, tolerance)
        old_owner = {i: r for r, o in enumerate(self.owned) for i in o}
        new_owner = {i: r for r, o in enumerate(new) for i in o}
        moves = sorted(i for i in new_owner if new_owner[i] != old_owner[i])
        received: dict = {}
        if moves and self.dist:
            send = [[i for i in moves if old_owner[i] == self.rank and new_owner[i] == r] for r in range(self.world)]
            recv = [[i for i in moves if new_owner[i] == self.rank and old_owner[i] == r] for r in range(self.world)]
            in_split = [sum(blob_size(i) for i in s) for s in send]
            out_split = [sum(blob_size(i) for i in s) for s in recv]
            parts = [pack(i).reshape(-1).to(device=self.device, dtype=torch.float32) for s in send for i in s]
            inp = torch.cat(parts) if parts else torch.empty(0, dtype=torch.float32, device=self.device)
            out = torch.empty(sum(out_split), dtype=torch.float32, device=self.device)
            self.dist.all_to_all_single(out, inp, out_split, in_split)
            self.collectives += 1
            off = 0
            for s in recv:
                for i in s:
                    received[i] = out[off:off + blob_size(i)]
                    off += blob_size(i)
        self.owned = new
        return new, received

    # ------------------------------------------------------------------ results back
    def gather(self, local: Sequence[torch.Tensor], counts: Optional[Sequence[int]] = None, dtype=torch.float32,
               wire_dtype: Optional[torch.dtype] = None) -> Optional[List[torch.Tensor]]:
        """Per-item 1-D results of this rank's items (in ``my_items()`` order) -> list over ALL items on rank ``dst`` (``None`` elsewhere).

        ``counts``: element count of every item (all ranks must pass the same list, e.g. frames x samples-per-frame); when omitted the counts are
        exchanged first (``share_counts``: one extra tiny all_reduce).  ONE all_to_all_single with exact split sizes carries the payload;
        ``wire_dtype`` (fp16 for waveforms) halves it and is cast back to ``dtype`` on ``dst``."""
        mine = self.my_items()
        assert len(local) == len(mine), (len(local), len(mine))
        if not self.dist:
            out: List[Optional[torch.Tensor]] = [None] * self.n_items
            for i, a in zip(mine, local):
                out[i] = a
            return out  # type: ignore[return-value]
        if counts is None:
            counts = self.share_counts([int(a.numel()) for a in local])
        wire = wire_dtype or dtype
        per_rank = [sum(int(counts[i]) for i in o) for o in self.owned]
        parts = [a.reshape(-1).to(device=self.device, dtype=wire) for a in local]
        inp = torch.cat(parts) if parts else torch.empty(0, dtype=wire, device=self.device)
        assert int(inp.numel()) == per_rank[self.rank], (int(inp.numel()), per_rank[self.rank])
        in_split = [per_rank[self.rank] if r == self.dst else 0 for r in range(self.world)]
        out_split = per_rank if self.rank == self.dst else [0] * self.world
        buf = torch.empty(sum(out_split), dtype=wire, device=self.device)
        self.dist.all_to_all_single(buf, inp, out_split, in_split)
        self.collectives += 1
        if self.rank != self.dst:
            return None
        res: List[Optional[torch.Tensor]] = [None] * self.n_items
        off = 0
        for o in self.owned:
            for i in o:
                n = int(counts[i])
                res[i] = buf[off:off + n].to(dtype)
                off += n
        return res  # type: ignore[return-value]


def scatter_dense(ch: ShardChannel, x: Optional[torch.Tensor], item_shape: Sequence[int]) -> Tuple[torch.Tensor, List[int]]:
    """Dense float32 requests (Whisper: log-mel windows) from rank ``src`` to every rank: ONE broadcast of a fixed-capacity buffer
    ``[max_items, 1 + prod(item_shape)]`` (column 0 = 1 for a valid row, -1 for an empty one).  Returns (this rank's items ``[n_mine, *item_shape]``,
    their global indices); every item costs the same, so the LPT plan is a round-robin by index.  The plan is stored in ``ch.owned``."""
    width = 1
    for d in item_shape:
        width *= int(d)
    buf = getattr(ch, "_dense_buf", None)
    if buf is None or buf.shape[1] != 1 + width:
        buf = torch.full((ch.max_items, 1 + width), -1.0, dtype=torch.float32, device=ch.device)
        ch._dense_buf = buf
    if ch.rank == ch.src:
        n = int(x.shape[0])
        if n > ch.max_items or tuple(x.shape[1:]) != tuple(int(d) for d in item_shape):
            raise ValueError(f"dense request batch {tuple(x.shape)} exceeds / mismatches the channel ({ch.max_items} items of {tuple(item_shape)})")
        buf[:, 0] = -1.0
        buf[:n, 0] = 1.0
        buf[:n, 1:] = x.reshape(n, -1).to(device=ch.device, dtype=torch.float32)
    if ch.dist:
        ch.dist.broadcast(buf, ch.src)
        ch.collectives += 1
        n = int((buf[:, 0] > 0).sum())
    ch.n_items = n
    ch.owned = lpt_assign([1] * n, ch.world)
    mine = ch.my_items()
    idx = torch.tensor(mine, dtype=torch.long, device=ch.device)
    return buf[idx, 1:].reshape(len(mine), *[int(d) for d in item_shape]), mine


def sharded_decode(ch: ShardChannel, requests: Optional[Sequence[torch.Tensor]], run_local: Callable[[List[int], List[torch.Tensor]], List[torch.Tensor]],
                   dtype: torch.dtype = torch.int64, gather: str = "rank0"):
    """The autoregressive configs (Qwen3-TTS, CSM, Whisper decode): sequences are independent, so a step is ``scatter_requests`` (one broadcast of
    the token block; the LPT plan on the prompt lengths) -> ``run_local(my global indices, my token ids)`` = the unchanged single-GPU engine on
    this rank's share -> the RAGGED results (one 1-D tensor per sequence: code frames, token ids, ...) back.
    ``gather="rank0"``: one tiny all_reduce of the result lengths + ONE exact-size all_to_all_single towards ``ch.dst`` (returns the list over all
    sequences there, ``None`` elsewhere) -- fine for code / token sequences (64 utterances x 48 frames x 16 codes x 8 B = 393 KB per step in
    total); ``gather="none"``: every rank keeps its own results (a serving shell streams each sequence from the rank that made it, rank 0 is no
    hot spot); returns ``(my indices, my results)``."""
    block, lens = ch.scatter_requests(requests)
    mine = ch.my_items()
    local = run_local(mine, ch.my_ids(block, lens))
    assert len(local) == len(mine)
    if gather == "none":
        return mine, local
    assert gather == "rank0", gather
    return ch.gather([a.reshape(-1) for a in local], dtype=dtype)


class ShardStepFailed(RuntimeError):
    """A sharded step failed on some rank.  Raised on EVERY rank at the same point of the protocol (after the collective that spread the news), so the
    group's collectives stay matched and the next step can run; ``__cause__`` holds the local exception on the rank(s) that failed."""


_FAILED_COUNT = -(1 << 28)   # a rank whose token-rate half raised reports this "frame count" for its items: the sum stays negative on every rank


def kokoro_step(ch: ShardChannel, engine, requests: Optional[Sequence[torch.Tensor]], ref_s_of: Callable,
                samples_per_frame: int, forced_durations_of: Optional[Callable[[int], torch.Tensor]] = None, tolerance: float = 0.05,
                wire_dtype: Optional[torch.dtype] = None, back_kwargs: Optional[Callable[[List[int]], dict]] = None, speed: Optional[float] = None):
    """One sharded Kokoro synthesis step: requests out, token-rate half on the token-LPT shard, frame counts shared, utterances re-balanced on
    the real frame counts, frame-rate half, waveforms back.  ``ref_s_of(item, n_tokens)`` -> the item's style row ``[1, 256]`` (or, when it has
    a ``rows`` attribute, ``ref_s_of.rows(items, lens)`` -> all rows ``[n, 256]`` in one device gather);
    ``back_kwargs(items)`` -> extra arguments of ``engine.back`` for the final shard (SineGen noise in the bench).  Returns the waveforms on
    ``ch.dst`` (``None`` elsewhere).  Collectives: broadcast, all_reduce (int32 counts), [all_to_all_single], all_to_all_single."""
    from .tts.models.kokoro.engine import KokoroFront

    block, lens = ch.scatter_requests(requests)
    first = ch.my_items()
    st = None
    err: Optional[BaseException] = None
    if first:
      try:   # a failure on this rank must not leave the others waiting in the next collective: it travels with the frame counts
        ids = ch.my_ids(block, lens)
        ref = ref_s_of.rows(first, lens) if hasattr(ref_s_of, "rows") else torch.cat([ref_s_of(i, lens[i]) for i in first], 0)
        kw = {}
        if hasattr(engine, "pb"):  # the real engine takes the batch as the block already holds it (rows of the request block, zero-padded)
            rows = torch.tensor(first, dtype=torch.long, device=block.device)
            kw["ids_padded"] = block[rows, 1:].clamp_min(0)
            if forced_durations_of is not None and hasattr(forced_durations_of, "rows"):
                kw["forced_padded"] = forced_durations_of.rows(first, lens)
        fd = [forced_durations_of(i) for i in first] if forced_durations_of and "forced_padded" not in kw else None
        if speed is not None:
            kw["speed"] = float(speed)
        st = engine.front(ids, ref, forced_durations=fd, **kw)
      except Exception as e:  # noqa: BLE001
        err, st = e, None
    frames = ch.share_counts(st.frames if st else [_FAILED_COUNT] * len(first if err else []))
    if any(f < 0 for f in frames):
        raise ShardStepFailed("the token-rate half of a sharded step failed on rank%s" % (" %d: %r" % (ch.rank, err) if err else " (another rank)")) from err
    width = engine.hid + engine.sty
    style = 2 * engine.sty
    pos = {i: k for k, i in enumerate(first)}
    new, received = ch.rebalance(frames, lambda i: st.pack(pos[i]), lambda i: KokoroFront.packed_size(lens[i], style, width), tolerance)
    mine = new[ch.rank]
    outs: List[torch.Tensor] = []
    if mine:
      try:
        if mine == first:
            merged = st  # nothing moved: the state goes on as it is (its padded device tensors included)
        else:
            kept = [i for i in mine if i in pos]
            parts = []
            if kept:
                parts.append((kept, st.select([pos[i] for i in kept])))
            got = [i for i in mine if i not in pos]
            if got:
                parts.append((got, KokoroFront.unpack([received[i] for i in got], [frames[i] for i in got], style, width, st.speed if st else 1.0)))
            order = [i for p in parts for i in p[0]]
            merged = KokoroFront.concat([p[1] for p in parts])
            perm = sorted(range(len(order)), key=lambda k: order[k])  # ascending item order == my_items() order
            merged = merged.select(perm)
        outs, _ = engine.back(merged, **(back_kwargs(mine) if back_kwargs else {}))
      except Exception as e:  # noqa: BLE001
        err, outs = e, []
    if ch.any_failed(err is not None):
        raise ShardStepFailed("the frame-rate half of a sharded step failed on rank%s" % (" %d: %r" % (ch.rank, err) if err else " (another rank)")) from err
    counts = [f * samples_per_frame for f in frames]
    return ch.gather(outs, counts=counts, wire_dtype=wire_dtype)


# ---------------------------------------------------------------------------------------------------- plain helpers (dense tensors)
def broadcast_tensor(t: Optional[torch.Tensor], device, dist=None, src: int = 0, dtype=torch.float32, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Rank ``src`` owns a dense tensor (e.g. the 30 s audio windows of a Whisper request, the prefill embeddings of a Qwen3 batch);
    afterwards every rank holds it on ``device``.  With ``shape`` given (the receivers know it: fixed window size) this is ONE broadcast;
    otherwise the shape travels first in one int64 header."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t.to(device=device, dtype=dtype)
    rank = dist.get_rank()
    if shape is None:
        hdr = torch.zeros(8, dtype=torch.int64, device=device)
        if rank == src:
            assert t.dim() <= 7
            hdr[0] = t.dim()
            for i, n in enumerate(t.shape):
                hdr[1 + i] = n
        dist.broadcast(hdr, src)
        hdr_h = hdr.cpu()
        shape = [int(v) for v in hdr_h[1:1 + int(hdr_h[0])]]
    buf = t.to(device=device, dtype=dtype).contiguous() if rank == src else torch.empty(list(shape), dtype=dtype, device=device)
    dist.broadcast(buf, src)
    return buf


# ---------------------------------------------------------------------------------------------------- the serving shell on N GPUs
class _RefRows:
    """``ref_s_of`` for ``kokoro_step`` when the style rows of the whole request batch are already on every rank."""

    def __init__(self, rows: torch.Tensor):
        self._rows = rows

    def __call__(self, i: int, n_tokens: int) -> torch.Tensor:
        return self._rows[i:i + 1]

    def rows(self, items: Sequence[int], lens: Sequence[int]) -> torch.Tensor:
        return self._rows[torch.tensor(list(items), dtype=torch.long, device=self._rows.device)]


class ShardedKokoro:
    """``engine.forward`` of the serving shell over every rank of a ``ShardChannel`` ("InferenceBroker / TTSBatchSession wired to the
    multi-GPU engine").  The HTTP front end, the ``InferenceBroker`` and the ``KokoroBatchSession`` live on rank ``ch.src`` and call ``forward`` exactly
    as they call the single-GPU engine; every other rank sits in ``worker_loop()``.  One call = one header broadcast (command, batch size, speed),
    one broadcast of the style rows ``[n, 2 * style]``, then ``kokoro_step`` (request block out, token-rate half on the token-LPT shard, frame counts
    shared, re-balance on the real frame counts, frame-rate half, waveforms back to ``ch.dst == ch.src``).  ``close()`` releases the workers."""

    CMD_STOP, CMD_RUN = 0, 1

    def __init__(self, engine, ch: ShardChannel, samples_per_frame: int, wire_dtype: Optional[torch.dtype] = None, tolerance: float = 0.05):
        assert ch.src == ch.dst, "the rank that submits the batch receives the waveforms"
        self.engine, self.ch, self.spf, self.wire_dtype, self.tolerance = engine, ch, int(samples_per_frame), wire_dtype, tolerance
        self.steps = 0

    def _header(self, cmd: int = 0, n: int = 0, speed: float = 1.0) -> Tuple[int, int, float]:
        ch = self.ch
        hdr = torch.zeros(3, dtype=torch.float64, device=ch.device)
        if ch.rank == ch.src:
            hdr = torch.tensor([float(cmd), float(n), float(speed)], dtype=torch.float64, device=ch.device)
        if ch.dist:
            ch.dist.broadcast(hdr, ch.src)
            ch.collectives += 1
        h = hdr.cpu()
        return int(h[0]), int(h[1]), float(h[2])

    def _run(self, input_ids, ref_s, n: int, speed: float):
        ch = self.ch
        width = 2 * self.engine.sty
        rows = broadcast_tensor(ref_s, ch.device, ch.dist, ch.src, shape=(n, width))
        if ch.dist:
            ch.collectives += 1
        self.steps += 1
        return kokoro_step(ch, self.engine, input_ids, _RefRows(rows), self.spf, tolerance=self.tolerance, wire_dtype=self.wire_dtype, speed=speed)

    # rank ch.src: the engine interface the sessions use
    def forward(self, input_ids: Sequence[torch.Tensor], ref_s: torch.Tensor, speed: float = 1.0, **kw):
        assert self.ch.rank == self.ch.src, "forward() is the submitting rank's call; the other ranks run worker_loop()"
        if kw:
            raise TypeError(f"ShardedKokoro.forward: unsupported arguments {sorted(kw)} (the sharded step takes ids, style rows and speed)")
        n = len(input_ids)
        # everything that can be checked is checked BEFORE the header goes out: once the workers have left their header wait they are inside the
        # step's collectives, and a rank-0 exception in front of those would strand them there
        ch = self.ch
        if n < 1 or n > ch.max_items:
            raise ValueError(f"ShardedKokoro.forward: {n} requests (the channel carries 1..{ch.max_items})")
        too_long = [int(t.numel()) for t in input_ids if int(t.numel()) > ch.max_tokens or int(t.numel()) < 1]
        if too_long:
            raise ValueError(f"ShardedKokoro.forward: request lengths {too_long} outside 1..{ch.max_tokens} tokens")
        width = 2 * self.engine.sty
        if ref_s.numel() != n * width:
            raise ValueError(f"ShardedKokoro.forward: ref_s must hold {n} style rows of {width} values, got shape {tuple(ref_s.shape)}")
        if not (speed > 0.0 and speed == speed and speed != float("inf")):
            raise ValueError(f"ShardedKokoro.forward: speed must be positive and finite (got {speed})")
        self._header(self.CMD_RUN, n, speed)
        outs = self._run(input_ids, ref_s.reshape(n, width), n, float(speed))   # ShardStepFailed: every rank left the step together; the group stays usable
        return outs, None

    def worker_loop(self) -> int:
        """Every rank but ``ch.src``: serve steps until ``close()``.  Returns the number of steps served."""
        assert self.ch.rank != self.ch.src
        while True:
            cmd, n, speed = self._header()
            if cmd == self.CMD_STOP:
                return self.steps
            try:
                self._run(None, None, n, speed)
            except ShardStepFailed:
                self.failed_steps = getattr(self, "failed_steps", 0) + 1   # reported to the caller on the submitting rank; this rank serves the next step

    def close(self) -> None:
        if self.ch.rank == self.ch.src:
            self._header(self.CMD_STOP)
