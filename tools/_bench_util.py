"""Helpers shared by the secondary benchmark lines (tools/bench_*.py)."""
import copy
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ev():
    return torch.cuda.Event(enable_timing=True)


def _clone_obj(o):
    """Deep copy of a layer record in which every device tensor is a fresh allocation (distinct HBM addresses)."""
    if isinstance(o, torch.Tensor):
        return o.clone()
    if dataclasses.is_dataclass(o):
        return type(o)(**{f.name: _clone_obj(getattr(o, f.name)) for f in dataclasses.fields(o)})
    if isinstance(o, tuple):
        return tuple(_clone_obj(v) for v in o)
    if isinstance(o, list):
        return [_clone_obj(v) for v in o]
    return copy.copy(o)


def build_deep_stack(cfg, device, seed=0, precision=2, weight_format="bf16", kv_dtype=torch.float32):
    """A TransformerStack of cfg.n_layers layers whose parameters are packed ONCE (one synthetic layer) and then cloned per layer on
    the device: identical values, distinct memory, so every layer streams its own weights from HBM exactly like a real checkpoint
    (host-side random generation + packing of >1e9 parameters would cost minutes of GPU-box time for nothing)."""
    from mlx_audio_amd.lm.stack import TransformerStack
    from mlx_audio_amd.lm.synthetic import make_stack_weights

    one = dataclasses.replace(cfg, n_layers=1)
    st = TransformerStack(make_stack_weights(one, seed=seed, gain=0.5), one, device=device, precision=precision, weight_format=weight_format,
                          kv_dtype=kv_dtype)
    st.cfg = cfg
    st.layers = [st.layers[0]] + [_clone_obj(st.layers[0]) for _ in range(cfg.n_layers - 1)]
    return st


def stack_weight_bytes(cfg, bytes_per_weight: float = 2.0) -> float:
    """Weight bytes one decode step of the stack streams (q|k|v, o, gate|up or w1, down or w2 of every layer): 2 per weight for the 16-bit
    images, 1 for fp8 (+ 4 per output row of scales)."""
    d, dh = cfg.d_model, cfg.head_dim
    per = (cfg.n_heads + 2 * cfg.n_kv_heads) * dh * d + d * cfg.n_heads * dh + (2 if cfg.mlp == "swiglu" else 1) * cfg.d_ff * d + d * cfg.d_ff
    rows = (cfg.n_heads + 2 * cfg.n_kv_heads) * dh + d + (2 if cfg.mlp == "swiglu" else 1) * cfg.d_ff + d
    return (bytes_per_weight * per + (4.0 * rows if bytes_per_weight < 2.0 else 0.0)) * cfg.n_layers


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def step_runner() -> str:
    """The decode-step runner behind mi355_stack_decode_step."""
    return "multi-launch native runner (stack_step.cpp)"


def cpu_frame_baseline(stacks, batch: int, context: int = 32, seconds_per_frame_audio: float = 0.08):
    """``cpu_baseline`` of a codec-LM frame loop: the oracle (oracle/lm_ref.py StackRef, PyTorch-CPU fp32, the restated reference) timed on this
    host for ONE generated frame of the same stacks -- ``stacks`` = [(StackConfig-like cfg, single-position steps per frame)].  Bounded sample:
    one layer's parameters are generated and aliased to every layer (generating > 1e9 random parameters on the host would cost minutes; the
    aliasing can only flatter the CPU through its caches), heads / embeddings / sampling are left out (also in the CPU's favour)."""
    import time
    from dataclasses import asdict

    from mlx_audio_amd.lm.synthetic import make_stack_weights
    from oracle import lm_ref as R

    cores = max(1, min(host_cores(), 32))
    torch.set_num_threads(cores)
    frame_s = 0.0
    for cfg, steps in stacks:
        one = dataclasses.replace(cfg, n_layers=1)
        w1 = make_stack_weights(one, seed=0)
        w = {}
        for k, v in w1.items():
            if k.startswith("layers.0."):
                for i in range(cfg.n_layers):
                    w[f"layers.{i}." + k[len("layers.0."):]] = v
            else:
                w[k] = v
        names = {f.name for f in dataclasses.fields(R.StackConfig)}
        ref = R.StackRef(w, R.StackConfig(**{k: v for k, v in asdict(cfg).items() if k in names}), param_dtype=torch.float32)
        cache = ref.make_cache()
        with torch.no_grad():
            ref(torch.randn(batch, context, cfg.d_model) * 0.1, cache)  # prefill (also the warm-up)
            n = max(1, min(steps, 4))
            t0 = time.perf_counter()
            for _ in range(n):
                ref(torch.randn(batch, 1, cfg.d_model) * 0.1, cache)
            frame_s += (time.perf_counter() - t0) / n * steps
        del ref, w, w1
    return {"value": batch * seconds_per_frame_audio / frame_s, "unit": "x realtime", "cores": cores, "kind": "port",
            "sample": "1 frame of the transformer stacks only (%s single-position steps, batch %d, context %d), oracle/lm_ref.py StackRef fp32; one "
                      "layer's random parameters aliased to all layers, heads / sampling / codec left out" % (
                          " + ".join(f"{s} x {c.n_layers} layers d={c.d_model}" for c, s in stacks), batch, context),
            "cpu_ms_per_frame": frame_s * 1e3}


class Dist:
    """One process per GPU (launched by ``python -m torch.distributed.run`` exactly like bench.py): RANK / LOCAL_RANK / WORLD_SIZE from the environment,
    backend "nccl" (= RCCL on ROCm).  A single process is world 1 with no process group."""

    def __init__(self):
        import os

        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.dev = torch.device("cuda", self.local)
        torch.cuda.set_device(self.dev)
        if self.world > 1:
            import torch.distributed as dist

            dist.init_process_group("nccl", device_id=self.dev)
            self.dist = dist

    def fence(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(self, v: float) -> float:
        if not self.dist:
            return v
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


class RehearsalEngine:
    """Stand-in with the Kokoro engine's ``front`` / ``back`` contract for ``bench.py --dry-run-gloo``: the multi-rank launch, the request broadcast,
    the frame-count all_reduce, the re-balance and the waveform all_to_all of mlx_audio_amd/shard.py are the REAL ones (over gloo, CPU tensors);
    only the arithmetic between them is replaced (frames = the forced durations' sum, 'waveform' = a ramp keyed on the state that crossed the
    split, so a wrong move would still be visible to a caller that checks it)."""
    hid, sty = 512, 128
    precision = 0

    def front(self, ids, ref_s, forced_durations=None, speed=1.0):
        from mlx_audio_amd.tts.models.kokoro.engine import KokoroFront

        ids = [i.to(torch.int32) for i in ids]
        d = [torch.zeros((int(i.numel()), self.hid + self.sty), dtype=torch.float32) + float(i.sum() % 7) for i in ids]
        dur = [f.to(torch.int32).reshape(-1)[: int(i.numel())] for f, i in zip(forced_durations, ids)]
        return KokoroFront(ids, ref_s.to(torch.float32), d, dur, [int(x.sum()) for x in dur], speed, None)

    def back(self, st, **kw):
        outs = [torch.arange(st.frames[b] * 600, dtype=torch.float32) * 1e-6 + float(st.d[b][0, 0]) for b in range(len(st.ids))]
        return outs, st.dur
