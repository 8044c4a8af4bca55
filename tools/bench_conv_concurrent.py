#!/usr/bin/env python
"""Do the three resblocks of an MRF stage (k = 3 / 7 / 11: HBM-bound / in between / matrix-pipe-bound) finish sooner side by side than one after the other?

    python tools/bench_conv_concurrent.py [--batch 32] [--out gpurun_out/conv_concurrent.txt]

One "round" = what a resblock triple does between two joins: conv k3, conv k7, conv k11 on independent tensors (Snake prologue with AdaIN
coefficients, fused statistics, residual), C = 128 at 120 F + 1 rows (stage 1) or C = 256 at 20 F rows (stage 0).
  sequential : the three launches on one stream, each on the whole chip (two persistent workgroups per CU) -- what the engine does today
  concurrent : three streams, each conv on a share of the persistent workgroups (mi355_conv_ws4_resident), shares from the sequential times or equal
Wall time per round from events on the main stream around fork ... join, median of --rounds.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--reps", type=int, default=6, help="rounds of the triple inside one timed region (the engine runs 6 per stage)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mlx_audio_amd import _lib, ops

    ops.require_gpu()
    lib = _lib.load()
    dev = "cuda"
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(0)
    lines = ["stage mode shares ms_per_round vs_sequential"]
    for C, L, dil in ((128, 31681, 1), (128, 31681, 5), (256, 5280, 1)):
        probs = []
        for k in (3, 7, 11):
            w = (torch.randn(C, k, C) / math.sqrt(k * C)).to(torch.bfloat16).float()
            pc = ops.pack_conv(w, torch.randn(C) * 0.1, dev)
            x = torch.randn((B, L, C), generator=g, device=dev)
            y = torch.zeros((B, L, C), device=dev)
            res = torch.randn((B, L, C), generator=g, device=dev)
            sc = torch.rand((B, C), generator=g, device=dev) + 0.5
            sh = torch.randn((B, C), generator=g, device=dev) * 0.3
            alpha = torch.rand(C, generator=g, device=dev) + 0.5
            st = ops.new_stats(B, L, C, dev)
            probs.append(dict(k=k, pc=pc, x=x, y=y, kw=dict(dil=dil, pad=(k * dil - dil) // 2, pre=(sc, sh), pre_act=ops.ACT_SNAKE, pre_alpha=alpha,
                                                              res=res if dil == 1 else None, stats=st, tile=6128128)))

        def run(p):
            ops.conv_gemm(p["x"], p["pc"], p["y"], **p["kw"])

        def timed(fn):
            ts = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / args.reps)
            return sorted(ts)[len(ts) // 2]

        # each conv alone (whole chip)
        alone = []
        for p in probs:
            run(p)
            alone.append(timed(lambda p=p: run(p)))
        seq = timed(lambda: [run(p) for p in probs])
        tag = f"C{C}_L{L}_d{dil}"
        lines.append(f"{tag} alone k3/k7/k11 {alone[0]:.4f}/{alone[1]:.4f}/{alone[2]:.4f} -")
        lines.append(f"{tag} sequential 512/512/512 {seq:.4f} 1.000")
        streams = [torch.cuda.Stream() for _ in probs]
        main_s = torch.cuda.current_stream()

        def concurrent(shares):
            ev = torch.cuda.Event()
            ev.record(main_s)
            for p, s, n in zip(probs, streams, shares):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    lib.mi355_conv_ws4_resident(int(n))
                    run(p)
                    done = torch.cuda.Event()
                    done.record(s)
                main_s.wait_event(done)
            lib.mi355_conv_ws4_resident(0)

        tot = sum(alone)
        prop = [max(8, int(round(512 * a / tot / 8)) * 8) for a in alone]
        for name, shares in (("prop", prop), ("equal", [168, 168, 176]), ("k11heavy", [96, 160, 256]), ("full", [512, 512, 512]), ("half", [256, 256, 256])):
            concurrent(shares)
            torch.cuda.synchronize()
            t = timed(lambda: concurrent(shares))
            lines.append(f"{tag} concurrent_{name} {'/'.join(str(s) for s in shares)} {t:.4f} {t / seq:.3f}")
        del probs
        torch.cuda.empty_cache()
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
