#!/usr/bin/env python
"""Phase timeline of the wave-specialised conv kernel (conv_ws4.hip, DBG instantiation, tile code 46128128): s_memtime stamps of consumer wave 0
and producer wave 4 of every 16th workgroup, reduced to medians per phase.

    python tools/conv_timeline.py [--batch 32] [--out gpurun_out/conv_timeline.txt]
"""
import argparse
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SLOTS = 48


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default="")
    ap.add_argument("--precision", type=int, default=2, help="2 (bf16 hi + lo) or 5 (fp16 hi + MX e4m3 lo: + barrier-wait sums and the producer wave's split)")
    args = ap.parse_args()
    from mlx_audio_amd import _lib, ops

    ops.require_gpu()
    lib = _lib.load()
    dev = "cuda"
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(0)
    lines = []
    for cin, cout, k, dil, L, res_on in ((128, 128, 11, 1, 31681, True), (128, 128, 11, 5, 31681, False), (128, 128, 3, 1, 31681, True),
                                         (128, 128, 7, 1, 31681, True), (256, 256, 7, 1, 5280, True)):
        w = (torch.randn(cout, k, cin) / math.sqrt(k * cin)).to(torch.bfloat16).float()
        if args.precision in (5, 6) and k % 4 != 3:
            continue
        pc = ops.pack_conv(w, torch.randn(cout) * 0.1, dev, mx=(args.precision - 4) if args.precision in (5, 6) else 0)
        x = torch.randn((B, L, cin), generator=g, device=dev)
        y = torch.zeros((B, L, cout), device=dev)
        sc = torch.rand((B, cin), generator=g, device=dev) + 0.5
        sh = torch.randn((B, cin), generator=g, device=dev) * 0.3
        alpha = torch.rand(cin, generator=g, device=dev) + 0.5
        res = torch.randn((B, L, cout), generator=g, device=dev) if res_on else None
        tiles = B * ((L + 127) // 128)
        grid = min(((tiles + 63) // 64) * 64 * ((cout + 127) // 128), 512)
        nrec = (grid + 15) // 16
        buf = torch.zeros((nrec, SLOTS), dtype=torch.int64, device=dev)
        kw = dict(dil=dil, pad=(k * dil - dil) // 2, pre=(sc, sh), pre_act=ops.ACT_SNAKE, pre_alpha=alpha, res=res, precision=args.precision)
        for tile in (46128128, 6128128, 46128128):  # probe once cold (code load), the plain kernel for the wall time, then the probe
            _lib.check(lib.mi355_conv_ws4_debug_buffer(ctypes.c_void_p(buf.data_ptr() if tile > 40000000 else 0)), "debug_buffer")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_gemm(x, pc, y, tile=tile, **kw)
            e1.record()
            torch.cuda.synchronize()
            if tile == 6128128:
                ms_plain = e0.elapsed_time(e1)
            ms = e0.elapsed_time(e1)
        _lib.check(lib.mi355_conv_ws4_debug_buffer(ctypes.c_void_p(0)), "debug_buffer")
        raw = buf.cpu().numpy().astype(np.int64)
        wc = raw[:, 32:34]
        t = raw[:, :32].reshape(nrec, 8, 4)
        okc = (wc[:, 1] > wc[:, 0]) & (t[:, 7, 0] > 0)
        ghz = float(np.median((t[okc, 7, 0] - t[okc, 0, 0]) / ((wc[okc, 1] - wc[okc, 0]) * 10.0))) if okc.any() else float("nan")  # [workgroup, tile, (start, window staged, main loop done, stores issued)]
        t = t[t[:, 0, 0] > 0]
        nch = (cin + 31) // 32
        mfma_cyc = k * 16 * 32 * nch if args.precision not in (5, 6) else (k * 8 * 32 + ((k + 1) // 2) * 4 * (64 if args.precision == 5 else 32)) * nch
        med = lambda v: float(np.median(v))
        lines.append(f"## cin={cin} cout={cout} k={k} dil={dil} rows={B * L} res={int(res_on)}: kernel {ms_plain * 1e3:.1f} us (probe build {ms * 1e3:.1f} us), grid {grid}, "
                     f"{len(t)} probed workgroups, shader clock {ghz:.2f} GHz (s_memtime ticks per wall_clock64 tick); one consumer wave issues {mfma_cyc} MFMA pipe cycles per tile (2 waves share a SIMD: {2 * mfma_cyc})")
        lines.append("tile#  wait-for-window  main-loop  epilogue  tile-period  barrier-wait(all chunks of the tile)   [shader cycles, medians over the probed workgroups]")
        bw = raw[raw[:, 0] > 0][:, 34:42]
        for i in range(8):
            ok = t[:, i, 3] > 0
            if not ok.any():
                break
            v = t[ok]
            period = med(v[:, i + 1, 0] - v[:, i, 0]) if i + 1 < 8 and (v[:, i + 1, 0] > 0).all() else float("nan")
            lines.append(f"{i:5d}  {med(v[:, i, 1] - v[:, i, 0]):15.0f}  {med(v[:, i, 2] - v[:, i, 1]):9.0f}  {med(v[:, i, 3] - v[:, i, 2]):8.0f}  {period:11.0f}  {med(bw[ok, i]):12.0f}")
        pr = raw[raw[:, 45] > 0][:, 42:48]
        if len(pr):
            life = pr[:, 5] - pr[:, 4]
            lines.append(f"producer wave 4 ({len(pr)} probed): items {med(pr[:, 3]):.0f}, life {med(life):.0f} cycles = convert {med(pr[:, 0] / life) * 100:.1f} % + barrier {med(pr[:, 1] / life) * 100:.1f} % + loads/bookkeeping {med(pr[:, 2] / life) * 100:.1f} %; "
                         f"per item: convert {med(pr[:, 0] / pr[:, 3]):.0f}, barrier {med(pr[:, 1] / pr[:, 3]):.0f}, loads {med(pr[:, 2] / pr[:, 3]):.0f} cycles")
        full = t[(t[:, 7, 3] > 0)]
        if len(full):
            per_tile = (full[:, 7, 3] - full[:, 0, 0]) / 8.0
            clk = None
            lines.append(f"8 tiles: {med(per_tile):.0f} cycles per tile and workgroup = {2 * mfma_cyc / med(per_tile) * 100:.0f} % of the SIMD's MFMA issue slots")
        del x, y, res
        torch.cuda.empty_cache()
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
