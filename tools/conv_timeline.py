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
        pc = ops.pack_conv(w, torch.randn(cout) * 0.1, dev)
        x = torch.randn((B, L, cin), generator=g, device=dev)
        y = torch.zeros((B, L, cout), device=dev)
        sc = torch.rand((B, cin), generator=g, device=dev) + 0.5
        sh = torch.randn((B, cin), generator=g, device=dev) * 0.3
        alpha = torch.rand(cin, generator=g, device=dev) + 0.5
        res = torch.randn((B, L, cout), generator=g, device=dev) if res_on else None
        tiles = B * ((L + 127) // 128)
        grid = ((tiles + 63) // 64) * 64 * ((cout + 127) // 128)
        nrec = (grid + 15) // 16
        buf = torch.zeros((nrec, 2, SLOTS), dtype=torch.int64, device=dev)
        kw = dict(dil=dil, pad=(k * dil - dil) // 2, pre=(sc, sh), pre_act=ops.ACT_SNAKE, pre_alpha=alpha, res=res)
        for tile in (6128128, 46128128):  # warm-up on the plain kernel, then the probe
            _lib.check(lib.mi355_conv_ws4_debug_buffer(ctypes.c_void_p(buf.data_ptr() if tile > 10000000 else 0)), "debug_buffer")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_gemm(x, pc, y, tile=tile, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        _lib.check(lib.mi355_conv_ws4_debug_buffer(ctypes.c_void_p(0)), "debug_buffer")
        t = buf.cpu().numpy().astype(np.int64)
        cons, prod = t[:, 0, :], t[:, 1, :]
        ok = (cons[:, 0] > 0) & (cons[:, SLOTS - 1] > 0) & (prod[:, 0] > 0)
        cons, prod = cons[ok], prod[ok]
        nch = (cin + 31) // 32
        span = int(cons[:, SLOTS - 1].max() - min(cons[:, 0].min(), prod[:, 0].min()))
        tick_ns = ms * 1e6 / span  # kernel wall time / stamp span
        med = lambda v: float(np.median(v)) * tick_ns / 1e3  # microseconds
        lines.append(f"## cin={cin} cout={cout} k={k} dil={dil} rows={B * L} res={int(res_on)}: kernel {ms * 1e3:.1f} us, {len(cons)} probed workgroups of {grid}, "
                     f"stamp span {span} ticks => {tick_ns:.3f} ns/tick")
        wg_total = cons[:, SLOTS - 1] - np.minimum(cons[:, 0], prod[:, 0])
        lines.append(f"workgroup lifetime (first stamp -> stores retired): median {med(wg_total):.2f} us  (p10 {med(np.percentile(wg_total, 10)):.2f}, p90 {med(np.percentile(wg_total, 90)):.2f})")
        lines.append(f"consumer: wait for window 0 (fold loads issued -> barrier #0 passed): {med(cons[:, 1] - cons[:, 0]):.2f} us")
        prev = cons[:, 1]
        comp, wait = [], []
        for c in range(nch - 1):
            bb, ba = cons[:, 2 + 2 * c], cons[:, 3 + 2 * c]
            comp.append(med(bb - prev)); wait.append(med(ba - bb))
            prev = ba
        comp.append(med(cons[:, SLOTS - 3] - prev))
        lines.append("consumer: MFMA time per chunk [us]: " + " ".join(f"{v:.2f}" for v in comp) + "   barrier wait after chunk: " + " ".join(f"{v:.2f}" for v in wait))
        lines.append(f"consumer: main loop total {med(cons[:, SLOTS - 3] - cons[:, 1]):.2f} us, epilogue (stores issued) {med(cons[:, SLOTS - 2] - cons[:, SLOTS - 3]):.2f} us, "
                     f"stores retired after {med(cons[:, SLOTS - 1] - cons[:, SLOTS - 2]):.2f} us")
        pl, pc_, pb = [], [], []
        prevp = prod[:, 0]
        for c in range(nch):
            a0, a1, a2 = prod[:, 1 + 3 * c], prod[:, 2 + 3 * c], prod[:, 3 + 3 * c]
            pl.append(med(a0 - prevp)); pc_.append(med(a1 - a0)); pb.append(med(a2 - a1))
            prevp = a2
        lines.append("producer: wait for loads per chunk [us]: " + " ".join(f"{v:.2f}" for v in pl))
        lines.append("producer: convert + issue next loads [us]:  " + " ".join(f"{v:.2f}" for v in pc_))
        lines.append("producer: barrier wait [us]:               " + " ".join(f"{v:.2f}" for v in pb))
        mfma_cyc = k * 16 * 32 * nch  # MFMA pipe cycles of one consumer wave for the tile
        lines.append(f"(one consumer wave issues {k * 16 * nch} MFMAs per tile = {mfma_cyc} pipe cycles = {mfma_cyc / 2.0 / 1e3:.2f} us at 2.0 GHz)")
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
