"""``BatchKVCache`` of the reference (``mlx_audio/lm/models/cache.py:502-717``) over the engine's KV layout.

Same semantics as the reference class -- left-padded rows, ``offset`` per row, step-256 growth with in-place slice update, ``trim``,
``filter`` / ``extend`` / ``extract`` / ``merge`` for continuous batching -- on one ``[B, capacity, 2 * n_kv * dh]`` float32 tensor per
layer (k columns first, then v; ``lm.stack.KVCache`` uses the same layout), so a cache built here plugs straight into
``mi355_flash_attention``: ``keys`` / ``values`` are column views, ``left_padding`` is the kernel's ``k_start`` and ``_idx`` its key count.
Pure tensor plumbing (no arithmetic on activations): works on any torch device, which is how the CPU tests exercise it.

Cost note: the reference re-merges every request's ``KVCache`` into a fresh ``BatchKVCache`` and extracts them again
on every continuous-batching step (continuous_batching.py:309-324) -- O(total KV bytes x 2) per frame per layer.  With this layout
``merge`` / ``extract`` are only needed when the set of requests changes; steps append in place.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .stack import KVCache


class BatchKVCache:
    step = 256

    def __init__(self, left_padding: Sequence[int], n_kv_heads: int, head_dim: int, device="cpu", dtype: torch.dtype = torch.float32):
        self.dtype = dtype
        self.width = 2 * n_kv_heads * head_dim
        self.n_kv_heads, self.head_dim = n_kv_heads, head_dim
        self.device = torch.device(device)
        self.kv: Optional[torch.Tensor] = None
        self.left_padding = torch.tensor(list(left_padding), dtype=torch.int32, device=self.device)
        self.offset = -self.left_padding.clone()
        self._idx = 0

    # ------------------------------------------------------------------ views the attention kernel consumes
    @property
    def keys(self):
        return None if self.kv is None else self.kv[:, :self._idx, : self.width // 2]

    @property
    def values(self):
        return None if self.kv is None else self.kv[:, :self._idx, self.width // 2:]

    @property
    def k_start(self) -> torch.Tensor:
        return self.left_padding

    def reserve(self, n_new: int) -> torch.Tensor:
        """update_and_fetch (cache.py:532-556) split in two: grow + advance here, the projection GEMM writes k | v into the returned slot."""
        B = self.left_padding.shape[0]
        prev = self._idx
        if self.kv is None or prev + n_new > self.kv.shape[1]:
            n_steps = (self.step + n_new - 1) // self.step
            new = torch.zeros((B, n_steps * self.step, self.width), dtype=self.dtype, device=self.device)
            if self.kv is not None:
                old = self.kv[:, :prev] if prev % self.step != 0 else self.kv
                self.kv = torch.cat([old, new], dim=1)
            else:
                self.kv = new
        self.offset = self.offset + n_new
        self._idx += n_new
        return self.kv[:, prev:self._idx, :]

    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor):
        """keys / values ``[B, n_new, n_kv * dh]`` (channels-last) -> (keys, values) views over everything cached so far."""
        slot = self.reserve(keys.shape[1])
        slot[:, :, : self.width // 2] = keys
        slot[:, :, self.width // 2:] = values
        return self.keys, self.values

    def size(self) -> int:
        return self._idx

    def empty(self) -> bool:
        return self.kv is None

    def is_trimmable(self) -> bool:
        return True

    def trim(self, n: int) -> int:
        n = min(self._idx, n)
        self._idx -= n
        self.offset = self.offset - n
        return n

    @property
    def nbytes(self) -> int:
        return 0 if self.kv is None else self.kv.numel() * self.kv.element_size()

    # ------------------------------------------------------------------ continuous batching (cache.py:606-717)
    def filter(self, batch_indices):
        idx = torch.as_tensor(batch_indices, dtype=torch.long, device=self.device)
        if self.kv is not None:
            self.kv = self.kv[idx]
        self.offset = self.offset[idx]
        self.left_padding = self.left_padding[idx]
        min_left = int(self.left_padding.min()) if self.left_padding.numel() else 0
        if min_left > 0:  # shift left to reduce padding
            if self.kv is not None:
                self.kv = self.kv[:, min_left:]
            self._idx -= min_left
            self.left_padding = self.left_padding - min_left

    def extend(self, other: "BatchKVCache"):
        if self.kv is None and other.kv is None:
            self.left_padding = torch.cat([self.left_padding, other.left_padding])
            self.offset = torch.cat([self.offset, other.offset])
            return
        max_idx = max(self._idx, other._idx)
        L1 = 0 if self.kv is None else self.kv.shape[1]
        L2 = 0 if other.kv is None else other.kv.shape[1]
        max_size = max(L1, L2)

        def pad(c: "BatchKVCache"):
            kv = c.kv
            if kv is None:
                kv = torch.zeros((c.offset.shape[0], 0, self.width), dtype=self.dtype, device=self.device)
            left = max_idx - c._idx
            right = max_size - kv.shape[1] - left
            if right < 0:
                kv = kv[:, :right]
                right = 0
            if left or right:
                kv = torch.nn.functional.pad(kv, (0, 0, left, right))
            return kv, c.offset, c.left_padding + left

        a, b = pad(self), pad(other)
        self.kv = torch.cat([a[0], b[0]], dim=0)
        self.offset = torch.cat([a[1], b[1]])
        self.left_padding = torch.cat([a[2], b[2]])
        self._idx = max_idx

    def extract(self, idx: int) -> KVCache:
        c = KVCache(self.n_kv_heads, self.head_dim, self.device, self.dtype)
        p = int(self.left_padding[idx])
        c.kv = self.kv[idx:idx + 1, p:self._idx].contiguous()
        c.offset = c.kv.shape[1]
        return c

    @classmethod
    def merge(cls, caches: List[KVCache]) -> "BatchKVCache":
        lengths = [c.size() for c in caches]
        max_length = max(lengths)
        ref = next((c for c in caches if c.kv is not None), caches[0])
        n_kv, dh, dev = ref.width // 2 // _dh(ref), _dh(ref), ref.device
        dt = getattr(ref, "dtype", torch.float32)
        if max_length == 0:
            return cls([0] * len(caches), n_kv, dh, dev, dt)
        padding = [max_length - l for l in lengths]
        out = cls(padding, n_kv, dh, dev, dt)
        out.kv = torch.zeros((len(caches), max_length, ref.width), dtype=dt, device=dev)
        for i, (p, c) in enumerate(zip(padding, caches)):
            if c.kv is not None:
                out.kv[i, p:p + c.offset] = c.kv[0, :c.offset]
        out.offset = out.offset + max_length
        out._idx = max_length
        return out


def _dh(c: KVCache) -> int:
    return getattr(c, "head_dim", None) or 64
