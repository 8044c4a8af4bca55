"""numpy / torch restatement of the reference's token sampling chain (TEST ORACLE, not product).

Follows ``tts/models/qwen3_tts/qwen3_tts.py:862-925`` (``_sample_token_batch``: suppress list -> per-sequence repetition penalty
over the set of generated tokens -> temperature -> top-k -> probability filters -> categorical) and ``lm/sample_utils.py``:
``apply_top_k`` :130-152, ``apply_min_p`` :155-198, ``apply_top_p`` :201-234, ``categorical_sampling`` :279-281.
``mx.random.categorical(logits)`` draws arg-max(logits + Gumbel noise); the noise is an explicit argument here so that two
implementations can be compared draw by draw.  Ties inside top-k (``mx.argpartition`` leaves them unspecified) resolve to the lower
index, like the device kernel.

Parity status: **pinned to the reference's own chain**: tests/golden/make_reference_fixtures.py ``run_sampler`` executes the reference's
``_sample_token_batch`` and ``lm/sample_utils.py`` (imported from /root/reference over the numpy stand-in for MLX) with the final categorical draw
replaced by a probe; ``ref_sampler.npz`` holds which logits survive the filters (and their values) for four parameter sets and the greedy arg-max;
tests/test_reference_fixtures_cpu.py::test_sampling_oracle_reproduces_the_reference_chain holds this file to them, and the same vectors feed the
device sampler's test.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

Tensor = torch.Tensor
NEG = -float("inf")


def apply_top_k(logprobs: Tensor, top_k: int) -> Tensor:
    V = logprobs.shape[-1]
    if not (0 < top_k < V):
        raise ValueError("top_k out of range")
    order = torch.argsort(-logprobs, dim=-1, stable=True)  # descending, stable => lower index first among ties
    out = logprobs.clone()
    out.scatter_(-1, order[..., top_k:], NEG)
    return out


def apply_top_p(logprobs: Tensor, top_p: float) -> Tensor:
    probs = torch.exp(logprobs)
    order = torch.argsort(logprobs, dim=-1, stable=True)  # ascending
    sp = torch.gather(probs, -1, order)
    cum = torch.cumsum(sp, dim=-1)
    inv = torch.empty_like(order)
    inv.scatter_(-1, order, torch.arange(order.shape[-1]).expand_as(order))
    cum = torch.gather(cum, -1, inv)
    return torch.where(cum > 1 - top_p, logprobs, torch.full_like(logprobs, NEG))


def apply_min_p(logprobs: Tensor, min_p: float) -> Tensor:
    top = logprobs.max(dim=-1, keepdim=True).values
    return torch.where(logprobs < top + math.log(min_p), torch.full_like(logprobs, NEG), logprobs)


def filter_logits(logits: Tensor, *, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.05,
                  generated: Optional[List[Sequence[int]]] = None, suppress_tokens: Optional[Sequence[int]] = None, min_p: float = 0.0) -> Tensor:
    """Everything of ``_sample_token_batch`` up to (not including) the categorical draw; logits [B, V] float32."""
    logits = logits.clone()
    if suppress_tokens:
        logits[:, list(suppress_tokens)] = NEG
    if generated and repetition_penalty != 1.0:
        for b, toks in enumerate(generated):
            valid = [t for t in set(toks) if t < logits.shape[-1]]
            if not valid:
                continue
            idx = torch.tensor(valid, dtype=torch.long)
            sel = logits[b, idx]
            logits[b, idx] = torch.where(sel < 0, sel * repetition_penalty, sel / repetition_penalty)
    if temperature <= 0:
        return logits
    if temperature != 1.0:
        logits = logits / temperature
    if 0 < top_k < logits.shape[-1]:
        logits = apply_top_k(logits, top_k)
    if 0.0 < top_p < 1.0 or min_p > 0.0:
        lp = torch.log_softmax(logits, dim=-1)
        if 0.0 < top_p < 1.0:
            lp = apply_top_p(lp, top_p)
        if min_p > 0.0:
            lp = apply_min_p(lp, min_p)
        logits = torch.where(lp == NEG, torch.full_like(logits, NEG), logits)
    return logits


def sample(logits: Tensor, gumbel: Optional[Tensor] = None, **kw) -> Tensor:
    f = filter_logits(logits, **kw)
    if kw.get("temperature", 0.9) <= 0 or gumbel is None:
        return f.argmax(dim=-1)
    return (f + gumbel).argmax(dim=-1)
