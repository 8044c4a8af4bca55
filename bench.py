#!/usr/bin/env python
"""Headline benchmark: audio samples/sec + real-time factor, Kokoro-82M TTS on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N>1 under
``python -m torch.distributed.run``, one rank per GPU, RCCL).  One "step" = one pass of the hot path
(tokens -> waveform, ``Model.__call__`` of the reference, kokoro.py:111-177) over one batch of
synthetic utterances per GPU: BASELINE.json config[1], "Kokoro-82M bf16 TTS on 1 MI355X".  Each
utterance is the canonical short sentence of BASELINE.md: 78 phonemes (T = 80 tokens), durations
forced to 264 frames -> 158 400 samples = 6.6 s at 24 kHz.  Weights: seeded random, bf16-valued, in
the exact Kokoro-82M shapes (no network => no checkpoint).  Inputs (token ids, voice rows, SineGen
noise) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES_PER_UTT = 158400
T_TOKENS, F_FRAMES = 80, 264
MFMA_BF16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA
HBM_PEAK_GBS = 8000.0            # same guide: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64,
                    help="utterances per GPU per step (SURVEY 8d batch list: 1 / 8 / 64 / 512; 64 per GPU = 512 over the 8-GPU node; measured on one\n"
                         "MI355X: 122 M samples/s at 32, 140 M at 64, 146 M at 128, 150 M at 256 -- the front end's latency-bound LSTMs amortise)")
    ap.add_argument("--precision", type=int, default=2,
                    help="2 = bf16 hi+lo split MFMA (fp32-grade), 3 = single fp16 pass in the vocoder, 1 = single bf16 pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shape-table", default="", help="write the per-shape conv_gemm timing table (roofline leg) to this file")
    return ap.parse_args()


def make_inputs(S, B, seed, dev):
    voice = S.make_voice_pack()
    ids = [S.make_phoneme_ids(T_TOKENS - 2, seed=seed * 1000 + i) for i in range(B)]
    ref_s = torch.cat([voice[T_TOKENS - 3] for _ in range(B)], 0)
    fds = [S.forced_durations(T_TOKENS, F_FRAMES, seed=seed * 1000 + i) for i in range(B)]
    rng = np.random.default_rng(1234 + seed)
    rand_ini = torch.from_numpy(rng.uniform(size=(B, 9)).astype(np.float32)).to(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    noise = torch.randn((B, 2 * F_FRAMES * 300, 9), generator=g, device=dev, dtype=torch.float32)
    return ids, ref_s.to(dev), fds, rand_ini, noise


def cpu_baseline(S):
    """The oracle (restated reference, PyTorch-CPU fp32) on this host's cores, bounded to ~10-30 s of CPU work:
    the canonical utterance if one pass takes < 8 s on this host, else a quarter-length one (F = 66)."""
    from oracle.kokoro_ref import KokoroRef

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))  # intra-op threads beyond ~32 only add synchronisation cost for these conv sizes
    torch.set_num_threads(cores)
    ref = KokoroRef(S.make_kokoro_weights(), S.KOKORO_CONFIG)
    ids = S.make_phoneme_ids(T_TOKENS - 2, seed=0)
    ref_s = S.make_voice_pack()[T_TOKENS - 3]

    def run(frames):
        fd = S.forced_durations(T_TOKENS, frames, seed=0)
        t0 = time.perf_counter()
        ref.forward(ids, ref_s, pred_dur=fd)
        return time.perf_counter() - t0

    probe = run(T_TOKENS)  # F = T = 80: also the warm-up
    frames = F_FRAMES if probe * F_FRAMES / T_TOKENS < 8.0 else 66
    reps = 3 if probe * frames / T_TOKENS < 4.0 else 1
    times = sorted(run(frames) for _ in range(reps))
    med = times[len(times) // 2]
    samples = frames * 600
    return {"value": samples / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 utterance (T=80, F={frames}, {samples} samples), 1 warm-up (F=80) + median of {reps}; restated "
                      "reference (oracle/kokoro_ref.py, PyTorch-CPU fp32), not MLX",
            "x_realtime": samples / 24000.0 / med, "host_cores_available": avail}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.gpus > 1:
        print("bench.py: --gpus > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from mlx_audio_amd import ops
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    from mlx_audio_amd import shard

    eng = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, device=dev, precision=args.precision)
    B = args.batch
    n_total = B * world
    # per-utterance inputs of THIS rank's shard are resident in HBM before the timed region; only the request batch
    # (token ids, owned by rank 0) and the waveforms cross ranks inside the step
    requests = [S.make_phoneme_ids(T_TOKENS - 2, seed=i) for i in range(n_total)] if rank == 0 else None
    _, lens0 = shard.broadcast_requests(requests, dev, dist)
    mine = shard.my_shard(lens0, dist)            # LPT over token counts: B utterances per rank here
    assert len(mine) == B, (len(mine), B)
    voice = S.make_voice_pack()
    ref_s = torch.cat([voice[T_TOKENS - 3] for _ in mine], 0).to(dev)
    fds = [S.forced_durations(T_TOKENS, F_FRAMES, seed=i) for i in mine]
    rng = np.random.default_rng(1234 + rank)
    rand_ini = torch.from_numpy(rng.uniform(size=(B, 9)).astype(np.float32)).to(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    noise = torch.randn((B, 2 * F_FRAMES * 300, 9), generator=g, device=dev, dtype=torch.float32)

    def step():
        # utterance sharding (mlx_audio_amd/shard.py): broadcast of the padded token batch from rank 0, local synthesis
        # of this rank's shard, gather of the waveforms on rank 0 -- over RCCL / xGMI when world > 1
        ids_pad, lens = shard.broadcast_requests(requests, dev, dist)
        local = [ids_pad[i, : T_TOKENS].long() for i in mine]
        outs, _ = eng.forward(local, ref_s, forced_durations=fds, rand_ini=rand_ini, noise=noise)
        shard.gather_waveforms(outs, mine, n_total, dev, dist)
        return outs

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(o.numel() == SAMPLES_PER_UTT for o in outs)
    assert all(bool(torch.isfinite(o).all()) for o in outs)

    res = None
    if rank == 0:
        total_samples = SAMPLES_PER_UTT * B * world * args.steps
        value = total_samples / dt
        res = {
            "metric": "audio samples/sec + real-time factor, Kokoro-82M TTS", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {2: "bf16 weights x fp32 activations (bf16 hi+lo split MFMA, fp32 accumulate)",
                      3: "fp16 activations x bf16-valued weights in fp16 (single MFMA pass, fp32 accumulate) in the vocoder; hi+lo front end",
                      1: "bf16"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "Kokoro-82M bf16 TTS, tokens->waveform, canonical short sentence T=80 F=264 (6.6 s @ 24 kHz)",
                       "utterances_per_gpu": B, "global_batch": B * world, "samples_per_utterance": SAMPLES_PER_UTT,
                       "parallelism": f"utterance-dp{world}", "precision_mode": args.precision},
            "x_realtime": value / 24000.0, "rtf_reference_style": 24000.0 / value,
        }
    # ---- roofline leg (rank 0, N=1 only): one extra instrumented step, events around every conv_gemm launch
    if rank == 0 and world == 1 and not args.no_roofline:
        ops.PROFILE = []
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.profile_finalize(ops.PROFILE), None
        if args.shape_table:
            agg = {}
            for fl, by, a0, a1, shp in prof:
                e = agg.setdefault(shp[:4], [0, 0.0, 0.0, 0])
                e[0] += 1; e[1] += a0.elapsed_time(a1); e[2] += fl; e[3] = shp[4]
            with open(args.shape_table, "w") as f:
                f.write("cin cout k dil rows launches ms gflop tflops\n")
                for shp, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write(f"{shp[0]} {shp[1]} {shp[2]} {shp[3]} {e[3]} {e[0]} {e[1]:.3f} {e[2] / 1e9:.2f} {e[2] / (e[1] * 1e-3) / 1e12:.1f}\n")
        flops = sum(p[0] for p in prof)
        byts = sum(p[1] for p in prof)
        # HBM bytes per conv_gemm launch from the PMC counters (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes over this same
        # command, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py writes the summary that is read back here)
        traffic, traffic_src = None, os.path.join(ROOT, "profiles", f"r1_hbm_traffic_kokoro_b{B}.json")
        if os.path.exists(traffic_src):
            with open(traffic_src) as f:
                pk = json.load(f)["kernels"]
            cg = [v for k, v in pk.items() if "conv_gemm" in k]
            if cg:
                traffic = sum(v["hbm_bytes_total_corrected"] for v in cg) / max(1, sum(v["dispatches"] for v in cg))
        ms = sum(p[2].elapsed_time(p[3]) for p in prof)
        res["roofline"] = {
            "bound": "mfma", "kernel": "conv_gemm_kernel (implicit-GEMM conv1d/convT/linear, v_mfma_f32_32x32x16_bf16)",
            "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
            "traffic_note": "avg HBM bytes per conv_gemm launch, PMC (2*FETCH_SIZE + WRITE_SIZE)*1024; algorithmic avg = %.3e B" % (byts / max(1, len(prof))),
            "launches_per_step": len(prof), "algorithmic_gflop_per_step": flops / 1e9,
            "conv_gemm_ms_per_step": ms, "instrumented_step_ms": e0.elapsed_time(e1),
            "hbm_view": {"algorithmic_GB_per_step": byts / 1e9, "achieved_GBps": byts / (ms * 1e-3) / 1e9,
                         "peak_GBps": HBM_PEAK_GBS, "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "note": "arithmetic intensity of the conv stack (~330-650 FLOP/B) is above the bf16 ridge (~312), so MFMA is the "
                    "binding roofline; the HBM view is reported alongside (DESIGN.md)",
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(S)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
