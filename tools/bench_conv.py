#!/usr/bin/env python
"""Within-process A/B of the conv_gemm tile variants on the Kokoro vocoder's dominant conv shapes.

    python tools/bench_conv.py [--batch 32] [--out gpurun_out/conv_ab.txt]

Every (shape, variant) pair is first checked against an fp64 torch conv of the same fused op on a
slice of the rows (the variants must agree with the reference, not merely with each other), then timed
interleaved over several rounds with events on the launch stream; the table reports the median.
Algorithmic FLOPs = 2*rows*Cin*Cout*K (one pass; the bf16 hi+lo split issues twice that on the MFMA
pipe), algorithmic bytes = fp32 in + out (+ residual) per row, weights once.
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--out", default="")
    ap.add_argument("--precision", type=int, default=2)
    ap.add_argument("--quick", action="store_true", help="two shapes, the wave-specialised variants only (for PMC passes)")
    ap.add_argument("--ablate", action="store_true", help="timing ablations of the wave-specialised kernel (their results are wrong by design)")
    ap.add_argument("--small", action="store_true", help="the few-row GEMMs of PL-BERT / predictor (rows = B*80): 4-wave tile shapes A/B")
    ap.add_argument("--prec-ab", action="store_true", help="the wave-specialised kernel under precisions 2 (bf16 hi+lo), 3 (one fp16 pass), 4 (fp16 hi+lo) and 5 (fp16 hi + MX e4m3 lo)")
    ap.add_argument("--pmc5", action="store_true", help="precision-5 wave-specialised kernel only, three shapes, few rounds (for rocprofv3 --pmc passes and power sampling)")
    ap.add_argument("--loop-seconds", type=float, default=0.0, help="with --pmc5: keep launching the first shape for this long (power / clock sampling by rocm-smi alongside)")
    ap.add_argument("--big-gemm", action="store_true", help="the Whisper-small encoder linears at 64 windows (96 000 rows): precisions 2 / 3 / 4 and the GEMM-mode timing ablations")
    ap.add_argument("--thin", action="store_true", help="the Kokoro generator's thin / shallow launches (22 -> 128 k = 1, 128 -> 22 k = 7, the flattened noise conv, the upsamplers): tile A/B")
    ap.add_argument("--flat", action="store_true", help="with --small: hand the batch over as ONE item of B*L rows (ops.conv_gemm(flatten=True))")
    args = ap.parse_args()
    from mlx_audio_amd import ops

    ops.require_gpu()
    dev = "cuda"
    B = args.batch
    # (cin, cout, k, dil, L, fused) -- stage-1 (C=128, L=31681) and stage-0 (C=256, L=5280) resblock convs
    shapes = []
    for k in (3, 7, 11):
        for dil in (1, 5):
            shapes.append((128, 128, k, dil, 31681, "snake+res" if dil == 1 else "snake"))
    for k in (3, 7, 11):
        shapes.append((256, 256, k, 1, 5280, "snake+res"))
    shapes.append((256, 256, 11, 5, 5280, "snake"))
    shapes.append((1090, 1024, 3, 1, 264, "leaky"))
    shapes.append((512, 2560, 2, 1, 529, "plain"))
    variants = [("old64x128", 64128), ("ws4_1tile", 86128128), ("ws4_masked_producers", 26128128), ("ws4", 6128128)]
    if args.ablate:
        shapes = [(128, 128, 11, 1, 31681, "snake+res"), (128, 128, 11, 5, 31681, "snake"), (128, 128, 3, 1, 31681, "snake+res"), (256, 256, 7, 1, 5280, "snake+res")]
        W = 6128128
        variants = [("ws4", W), ("abl1_noBload", 100000000 + W), ("abl2_noAread", 200000000 + W), ("abl3_noAB", 300000000 + W),
                    ("abl4_noProducer", 400000000 + W), ("abl7_noABP", 700000000 + W)]
        if args.precision in (5, 6):
            shapes = [(128, 128, 11, 1, 31681, "snake+res"), (128, 128, 7, 1, 31681, "snake+res"), (128, 128, 11, 5, 31681, "snake"), (256, 256, 11, 1, 5280, "snake+res")]
            shapes += [(128, 128, 11, 3, 31681, "snake"), (128, 128, 7, 5, 31681, "snake"), (256, 256, 7, 1, 5280, "snake+res"), (256, 256, 11, 5, 5280, "snake")]
            variants = [("ws4", W), ("ws4_2x2", 20000000 + W), ("abl1_noBload", 100000000 + W), ("abl4_noProducer", 400000000 + W), ("abl5_noBload_noProducer", 500000000 + W)]
            if args.precision == 6:
                variants = [v for v in variants if v[0] != "ws4_2x2"]   # the FP4 lo pass exists in the column-wave layout only
    if args.small:
        shapes = [(768, 2304, 1, 1, 80, "plain"), (768, 768, 1, 1, 80, "plain"), (768, 2048, 1, 1, 80, "plain"), (2048, 768, 1, 1, 80, "plain"),
                  (640, 2048, 1, 1, 264, "plain"), (1090, 1024, 3, 1, 264, "leaky"), (1024, 1024, 3, 1, 264, "leaky"), (512, 512, 3, 1, 528, "leaky"),
                  (512, 512, 5, 1, 80, "plain")]
        variants = [("t64x128", 64128), ("t64x64", 64064), ("ws4_1tile", 86128128), ("ws4", 6128128)]
    if args.thin:
        shapes = [(22, 128, 1, 1, 31681, "plain"), (128, 22, 7, 1, 31681, "leaky"), (288, 256, 1, 1, 5280, "plain"), (256, 768, 2, 1, 5281, "plain"), (512, 2560, 2, 1, 529, "plain")]
        variants = [("auto", 0), ("t64x64", 64064), ("t64x128", 64128), ("t128x128", 128128), ("ws4_n64", 6128064), ("ws4_n64_noprio", 16128064), ("ws4", 6128128), ("ws4_noprio", 16128128)]
    if args.big_gemm:
        W = 6128128
        shapes = [(768, 2304, 1, 1, 1500, "plain"), (768, 768, 1, 1, 1500, "plain+res"), (768, 3072, 1, 1, 1500, "plain"), (3072, 768, 1, 1, 1500, "plain+res")]
        variants = [("p2_bf16_hi_lo", W), ("p3_fp16_one_pass", W), ("p4_fp16_hi_lo", W), ("p2_masked_producers", 20000000 + W), ("p4_masked_producers", 20000000 + W), ("p4_noprio", 10000000 + W), ("p4_1tile_per_wg", 80000000 + W), ("p2_xsplit", W), ("p4_xsplit", W), ("p4_xsplit_ysplit", W),
                    ("p2_abl1_noBload", 100000000 + W), ("p2_abl4_noProducer", 400000000 + W), ("p2_abl5_noBload_noProducer", 500000000 + W)]
    if args.quick:
        shapes = [(128, 128, 11, 1, 31681, "snake+res"), (128, 128, 3, 1, 31681, "snake+res")]
        variants = [("old64x128", 64128), ("ws4", 6128128)]
    if args.pmc5:
        args.precision = 5
        shapes = [(128, 128, 11, 1, 31681, "snake+res"), (128, 128, 7, 1, 31681, "snake+res"), (256, 256, 11, 1, 5280, "snake+res")]
        variants = [("ws4", 6128128)]
        args.rounds = 3
    if args.prec_ab:
        shapes = [s for s in shapes if s[2] % 4 == 3]
        variants = [("p2_bf16_hi_lo", 6128128), ("p3_fp16_one_pass", 6128128), ("p4_fp16_hi_lo", 6128128), ("p5_fp16_hi_mx8_lo", 6128128), ("p6_fp16_hi_mx4_lo", 6128128)]
    lines = ["cin cout k dil rows fused variant ms tflops_alg GBps_alg maxrel"]
    g = torch.Generator(device=dev).manual_seed(0)
    for cin, cout, k, dil, L, fused in shapes:
        if args.flat:
            L, B = L * args.batch, 1
        w = (torch.randn(cout, k, cin) / math.sqrt(k * cin)).to(torch.bfloat16).float()
        bias = torch.randn(cout) * 0.1
        pc = ops.pack_conv(w, bias, dev, mx=args.precision - 4) if args.precision in (5, 6) else ops.pack_conv(w, bias, dev, f16=args.precision == 3)
        vprec = {name: args.precision for name, _ in variants}
        vpc = {name: pc for name, _ in variants}
        if args.big_gemm:
            pc16 = ops.pack_conv(w, bias, dev, f16=True)
            vprec = {name: (3 if name.startswith("p3") else 4 if name.startswith("p4") else 2) for name, _ in variants}
            vpc = {name: (pc16 if vprec[name] != 2 else pc) for name, _ in variants}
        if args.prec_ab:
            pc16, pcmx, pcmx4 = ops.pack_conv(w, bias, dev, f16=True), ops.pack_conv(w, bias, dev, mx=True), ops.pack_conv(w, bias, dev, mx=2)
            vprec = {"p2_bf16_hi_lo": 2, "p3_fp16_one_pass": 3, "p4_fp16_hi_lo": 4, "p5_fp16_hi_mx8_lo": 5, "p6_fp16_hi_mx4_lo": 6}
            vpc = {"p2_bf16_hi_lo": pc, "p3_fp16_one_pass": pc16, "p4_fp16_hi_lo": pc16, "p5_fp16_hi_mx8_lo": pcmx, "p6_fp16_hi_mx4_lo": pcmx4}
        ld = ops.round_up(cin, 32)
        x = torch.randn((B, L, ld), generator=g, device=dev)
        y = torch.zeros((B, L, cout), device=dev)
        pad = (k * dil - dil) // 2
        kw = {}
        if fused.startswith("snake"):
            sc = torch.rand((B, ld), generator=g, device=dev) + 0.5
            sh = torch.randn((B, ld), generator=g, device=dev) * 0.3
            alpha = torch.rand(ld, generator=g, device=dev) + 0.5
            kw.update(pre=(sc, sh), pre_act=ops.ACT_SNAKE, pre_alpha=alpha)
        elif fused == "leaky":
            sc = torch.rand((B, ld), generator=g, device=dev) + 0.5
            sh = torch.randn((B, ld), generator=g, device=dev) * 0.3
            kw.update(pre=(sc, sh), pre_act=ops.ACT_LEAKY, pre_slope=0.2)
        res = None
        if fused.endswith("+res"):
            res = torch.randn((B, L, cout), generator=g, device=dev)
            kw.update(res=res)
        rows = B * L
        flops = 2.0 * rows * cin * cout * k
        byts = 4.0 * rows * (cin + cout + (cout if res is not None else 0)) + 2.0 * cin * cout * k
        # reference on the first item's first rows
        n = min(L, 700)
        xr = x[0:1, :n, :cin].double().cpu()
        if "pre" in kw:
            xr = xr * kw["pre"][0][0, :cin].double().cpu() + kw["pre"][1][0, :cin].double().cpu()
            if kw["pre_act"] == ops.ACT_SNAKE:
                al = kw["pre_alpha"][:cin].double().cpu()
                xr = xr + (1.0 / al) * torch.sin(al * xr) ** 2
            else:
                xr = F.leaky_relu(xr, 0.2)
        ref = F.conv1d(F.pad(xr.transpose(1, 2), (pad, (k - 1) * dil - pad)), w.permute(0, 2, 1).double(), bias.double(),
                       dilation=dil).transpose(1, 2)[0]
        if res is not None:
            ref = ref + res[0, :n].double().cpu()
        ok_rows = n - (k - 1) * dil  # rows whose window stays inside the first n inputs
        times = {name: [] for name, _ in variants}
        errs = {}
        xsplit = {}   # pre-split copies of the input (the split variants: conversion done once, in front of the launch)
        for name, _ in variants:
            if "xsplit" in name and vprec[name] not in xsplit:
                xsplit[vprec[name]] = ops.split16(x, vprec[name])

        def xin(name):
            return (xsplit[vprec[name]] if "xsplit" in name else x)[:, :, :cin]

        def vkw(name):
            return dict(kw, x_split="xsplit" in name, y_split="ysplit" in name) if "split" in name else kw

        for name, tile in variants:
            try:
                ops.conv_gemm(xin(name), vpc[name], y, dil=dil, pad=pad, tile=tile, precision=vprec[name], **vkw(name))
                torch.cuda.synchronize()
                if "ysplit" in name:   # the stored words' hi + lo halves against the reference
                    wds = y[0, :ok_rows].contiguous().view(torch.int32)
                    halves = torch.stack([wds & 0xffff, (wds >> 16) & 0xffff], 0).to(torch.int16)
                    hl = halves.view(torch.float16 if vprec[name] == 4 else torch.bfloat16).double().cpu()
                    got = hl[0] + hl[1]
                else:
                    got = y[0, :ok_rows].double().cpu()
                errs[name] = float((got - ref[:ok_rows]).abs().max() / ref.abs().max())
            except Exception as e:  # a variant may not support a shape
                errs[name] = str(e)[:60]
        if args.loop_seconds > 0 and (cin, k) == (shapes[0][0], shapes[0][2]):
            import time
            t_end = time.time() + args.loop_seconds
            n = 0
            while time.time() < t_end:
                for _ in range(50):
                    ops.conv_gemm(x[:, :, :cin], vpc["ws4"], y, dil=dil, pad=pad, tile=6128128, precision=vprec["ws4"], **kw)
                torch.cuda.synchronize()
                n += 50
            print(f"# sustained loop: {n} launches in {args.loop_seconds:.1f} s = {args.loop_seconds / n * 1e6:.1f} us per launch", flush=True)
        for _ in range(args.rounds):
            for name, tile in variants:
                if isinstance(errs[name], str):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv_gemm(xin(name), vpc[name], y, dil=dil, pad=pad, tile=tile, precision=vprec[name], **vkw(name))
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1))
        for name, _ in variants:
            if isinstance(errs[name], str):
                lines.append(f"{cin} {cout} {k} {dil} {rows} {fused} {name} n/a - - {errs[name]!r}")
                continue
            ms = sorted(times[name])[len(times[name]) // 2]
            lines.append(f"{cin} {cout} {k} {dil} {rows} {fused} {name} {ms:.4f} {flops / ms / 1e9:.1f} {byts / ms / 1e6:.0f} {errs[name]:.2e}")
        del x, y, res, xsplit
        torch.cuda.empty_cache()
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
