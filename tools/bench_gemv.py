#!/usr/bin/env python
"""Launch-period microbenchmark of mi355_gemv on the decode-step shapes of the three autoregressive configs (Whisper-small decoder, Qwen3-TTS-1.7B
talker / code predictor, CSM-1B backbone / depth decoder): N back-to-back launches of one shape on one stream (each waits for its predecessor, like the
dependent chain of a decode step), events around the batch.  Prints one line per shape: us per launch, weight GB/s, fraction of the 8 TB/s HBM peak.
A/B knobs are environment variables read by the library (MI355_GEMV_TWO_READS=1: the older fused-norm schedule)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (label, M, N, K, norm, glu, f16)
    ("whisper qkv", 8, 2304, 768, "layer", False, True), ("whisper out", 8, 768, 768, None, False, True), ("whisper mlp1", 8, 3072, 768, "layer", False, True),
    ("whisper mlp2", 8, 768, 3072, None, False, True), ("whisper logits", 8, 51865, 768, "layer", False, True),
    ("talker qkv", 8, 4096, 2048, "rms", False, False), ("talker wo", 8, 2048, 2048, None, False, False), ("talker gate|up", 8, 12288, 2048, "rms", True, False),
    ("talker down", 8, 2048, 6144, None, False, False),
    ("codepred qkv", 8, 4096, 1024, "rms", False, False), ("codepred wo", 8, 1024, 2048, None, False, False), ("codepred gate|up", 8, 6144, 1024, "rms", True, False),
    ("codepred down", 8, 1024, 3072, None, False, False),
    ("csm bb qkv", 1, 3072, 2048, "rms", False, False), ("csm bb wo", 1, 2048, 2048, None, False, False), ("csm bb gate|up", 1, 16384, 2048, "rms", True, False),
    ("csm bb down", 1, 2048, 8192, None, False, False),
    ("csm dec qkv", 1, 1536, 1024, "rms", False, False), ("csm dec wo", 1, 1024, 1024, None, False, False), ("csm dec gate|up", 1, 16384, 1024, "rms", True, False),
    ("csm dec down", 1, 1024, 8192, None, False, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--tag", default="")
    ap.add_argument("--graph", action="store_true", help="replay the launches from a HIP graph (kernels shorter than the ~12 us host cost of a ctypes launch)")
    ap.add_argument("--pipe", action="store_true", help="rows pipeline (rows_pipe.hip) instead of mi355_gemv: converter, GEMM and row epilogue timed separately")
    ap.add_argument("--only", default="", help="substring filter on the shape label")
    ap.add_argument("--rows", type=int, default=0, help="override M of every Qwen3 / Whisper shape (9..64: the gemm_rows.hip kernel)")
    args = ap.parse_args()
    from mlx_audio_amd import ops

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    rows = []
    for label, M, N, K, norm, glu, f16 in SHAPES:
        if args.only and args.only not in label:
            continue
        if args.rows:
            if label.startswith("csm"):
                continue
            M = args.rows
        w = (torch.randn(N, K, generator=g) / K ** 0.5)
        w = w.half().float() if f16 else w.bfloat16().float()
        rw = ops.pack_rowmajor16(w, torch.zeros(N), dev, f16=f16)
        x = torch.randn(M, K, generator=g).to(dev)
        y = torch.empty(M, N // 2 if glu else N, device=dev)
        nrm = None
        if norm:
            nrm = (norm, torch.ones(K, device=dev), torch.zeros(K, device=dev) if norm == "layer" else None, 1e-5)
        if args.pipe:
            tl = ops.tiles16_from_rowmajor(rw)
            R = ops.rows_R(M)
            planes = ops.rows_planes(R, K, dev)
            kg = ops.rows_kgroups(N, K)
            part = torch.empty(kg, M, (N + 7) // 8 * 8, device=dev)
            po = ops.rows_planes(R, N // 2 if glu else N, dev) if (N // 2 if glu else N) % 64 == 0 and (N // 2 if glu else N) <= 8192 else None
            stages = {"convert": lambda: ops.rows_finish(x, M, K, norm=nrm, planes=planes, R=R, f16=f16),
                      "gemm": lambda: ops.rows_gemm(planes, tl, part, M, R, kgroups=kg),
                      "finish": lambda: ops.rows_finish(part, M, N, kg, bias=rw.bias, glu=glu, y=None if po is not None else y, planes=po, R=R, f16=f16)}
            res = {}
            for name, fn in stages.items():
                fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                s_ = torch.cuda.Stream()
                with torch.cuda.stream(s_):
                    fn()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gr, stream=s_):
                        for _ in range(args.iters):
                            fn()
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) * 1000.0 / args.iters
            gbs = 2.0 * N * K / (res["gemm"] * 1e-6) / 1e9
            rows.append(dict(shape=label, M=M, N=N, K=K, kgroups=kg, us_convert=round(res["convert"], 2), us_gemm=round(res["gemm"], 2), us_finish=round(res["finish"], 2),
                             GBps_gemm=round(gbs, 1), frac_hbm_gemm=round(gbs / 8000.0, 4)))
            print(f"{args.tag:10s} {label:18s} M={M} N={N:6d} K={K:5d} kg={kg:2d}  convert {res['convert']:6.2f}  gemm {res['gemm']:7.2f} us ({gbs:7.1f} GB/s {gbs / 8000.0:.3f})  finish {res['finish']:6.2f}")
            continue
        for _ in range(20):
            ops.gemv(x, rw, y, glu=glu, norm=nrm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.graph:
            gr = torch.cuda.CUDAGraph()
            s_ = torch.cuda.Stream()
            with torch.cuda.stream(s_):
                ops.gemv(x, rw, y, glu=glu, norm=nrm)
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s_):
                    for _ in range(args.iters):
                        ops.gemv(x, rw, y, glu=glu, norm=nrm)
            gr.replay()
            torch.cuda.synchronize()
            e0.record()
            gr.replay()
            e1.record()
        else:
            e0.record()
            for _ in range(args.iters):
                ops.gemv(x, rw, y, glu=glu, norm=nrm)
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / args.iters
        gbs = 2.0 * N * K / (us * 1e-6) / 1e9
        rows.append(dict(shape=label, M=M, N=N, K=K, norm=norm or "-", glu=glu, us=round(us, 2), GBps=round(gbs, 1), frac_hbm=round(gbs / 8000.0, 4)))
        print(f"{args.tag:10s} {label:18s} M={M} N={N:6d} K={K:5d} norm={norm or '-':5s} glu={int(glu)}  {us:8.2f} us  {gbs:8.1f} GB/s  {gbs / 8000.0:.3f}")
    print(json.dumps({"tag": args.tag, "rows": rows}))


if __name__ == "__main__":
    main()
