#!/usr/bin/env python
"""Diagnostic: ONE conv launch with fused statistics on B identical rows (the shape that first went row-dependent in tools/diag_batch_ops.py):
where do the statistics partials of the rows differ, and do they equal the statistics of the stored output?

    python tools/diag_conv_stats.py [--batch 64] [--L 264] [--C 512] [--k 3] [--tile 0] [--precision 2] [--act leaky]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--L", type=int, default=264)
    ap.add_argument("--C", type=int, default=512)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--precision", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--nolens", action="store_true")
    args = ap.parse_args()
    from mlx_audio_amd import ops

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    B, L, C, K = args.batch, args.L, args.C, args.k
    w = (torch.randn(C, K, C, generator=g) / math.sqrt(K * C)).bfloat16().float()
    bias = torch.randn(C, generator=g) * 0.1
    pc = ops.pack_conv(w, bias, dev, mx=args.precision == 5)
    x1 = torch.randn(1, L, C, generator=g)
    x = x1.expand(B, -1, -1).contiguous().to(dev)
    sc = (torch.rand(1, C, generator=g) + 0.5).expand(B, -1).contiguous().to(dev)
    sh = (torch.randn(1, C, generator=g) * 0.1).expand(B, -1).contiguous().to(dev)
    lens = None if args.nolens else torch.full((B,), L, dtype=torch.int32, device=dev)
    nblk = (L + 63) // 64
    for rep in range(args.reps):
        y = torch.zeros(B, L, C, device=dev)
        st = ops.new_stats(B, L, C, dev)
        st.fill_(float("nan"))
        kw = dict(pad=(K - 1) // 2, lens_in=lens, lens_out=lens, pre=(sc, sh), pre_act=ops.ACT_LEAKY, pre_slope=0.2, stats=st, precision=args.precision)
        if args.tile:
            kw["tile"] = args.tile
        ops.conv_gemm(x, pc, y, **kw)
        torch.cuda.synchronize()
        yd = float((y - y[0:1]).abs().max())
        nan = int(torch.isnan(st).sum())
        # reference partials from the stored output
        ref = torch.empty_like(st)
        for e in range(nblk):
            blk = y[:, e * 64:min(L, (e + 1) * 64)].double()
            ref[:, e, :, 0] = blk.sum(1).float()
            ref[:, e, :, 1] = ((blk - blk.mean(1, keepdim=True)) ** 2).sum(1).float()
        d = (st - ref).abs()
        d[torch.isnan(d)] = float("inf")
        tol = 1e-3 * ref.abs().amax()
        bad = (d > tol)
        print(f"rep {rep}: y row spread {yd:.2e}; stats NaN entries (never written) {nan}; entries off by > {float(tol):.2e}: {int(bad.sum())} of {bad.numel()}")
        if int(bad.sum()):
            idx = bad.nonzero()
            bs = sorted(set(int(i[0]) for i in idx))
            blks = sorted(set(int(i[1]) for i in idx))
            cols = sorted(set(int(i[2]) for i in idx))
            print(f"   batch rows {bs[:20]}{'...' if len(bs) > 20 else ''} ({len(bs)}); blocks {blks}; columns {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)})")
            for i in idx[:6]:
                b, e, n, c2 = (int(v) for v in i)
                print(f"   st[{b},{e},{n},{c2}] = {float(st[b, e, n, c2]):.5f}   from y: {float(ref[b, e, n, c2]):.5f}   row 0's: {float(st[0, e, n, c2]):.5f}")


if __name__ == "__main__":
    main()
