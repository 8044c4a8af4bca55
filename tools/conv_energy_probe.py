#!/usr/bin/env python
"""Is conv_ws4 precision 5 bound by stalls or by the power cap?  One shape, the same instruction stream, with the HBM traffic / the data activity removed:

    normal        distinct [B, L, C] tensors (the contract workload's conv)
    alias_in      x and res of every utterance alias utterance 0 (batch stride 0: the reads stay in L2 / the Infinity Cache), y distinct
    alias_all     y aliased too: no HBM traffic to speak of
    zero_act      AdaIN scale = shift = 0: every prologue value is 0 (Snake(0) = 0) -> the matrix pipe multiplies zeros (toggle power gone, same instructions)
    no_res        no residual operand

Each case runs back to back for --seconds while rocm-smi is sampled alongside (socket power, sclk); the table says what each removal buys.
    python tools/conv_energy_probe.py [--batch 64] [--seconds 2.0] [--out gpurun_out/conv_energy_probe.txt]
"""
import argparse
import math
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.samples = False, []

    def run(self):
        while not self.stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
                p = re.search(r"Package Power \(W\):\s*([\d.]+)", o)
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                if p and c:
                    self.samples.append((float(p.group(1)), float(c.group(1))))
            except Exception:
                pass
            time.sleep(0.15)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mlx_audio_amd import ops

    ops.require_gpu()
    dev = "cuda"
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(0)
    lines = ["cin k dil rows case us_per_launch tflops_alg power_W sclk_MHz"]
    for cin, k, dil, L in ((128, 11, 1, 31681), (128, 7, 1, 31681), (128, 3, 1, 31681), (256, 11, 1, 5280)):
        cout = cin
        w = (torch.randn(cout, k, cin) / math.sqrt(k * cin)).to(torch.bfloat16).float()
        pc = ops.pack_conv(w, torch.randn(cout) * 0.1, dev, mx=True)
        x = torch.randn((B, L, cin), generator=g, device=dev)
        res = torch.randn((B, L, cout), generator=g, device=dev)
        y = torch.zeros((B, L, cout), device=dev)
        sc = torch.rand((B, cin), generator=g, device=dev) + 0.5
        sh = torch.randn((B, cin), generator=g, device=dev) * 0.3
        alpha = torch.rand(cin, generator=g, device=dev) + 0.5
        al = lambda t: t[0:1].expand(B, -1, -1)
        zs = torch.zeros_like(sc)
        cases = [("normal", x, res, y, sc, sh), ("alias_in", al(x), al(res), y, sc, sh), ("alias_all", al(x), al(res), al(y), sc, sh),
                 ("zero_act", x, res, y, zs, zs), ("no_res", x, None, y, sc, sh), ("normal_again", x, res, y, sc, sh)]
        flops = 2.0 * B * L * cin * cout * k
        for name, xx, rr, yy, s0, s1 in cases:
            kw = dict(dil=dil, pad=(k * dil - dil) // 2, pre=(s0, s1), pre_act=ops.ACT_SNAKE, pre_alpha=alpha, res=rr, precision=5, tile=6128128)
            for _ in range(3):
                ops.conv_gemm(xx, pc, yy, **kw)
            torch.cuda.synchronize()
            smi = Smi()
            smi.start()
            t0 = time.time()
            n = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.time() - t0 < args.seconds:
                for _ in range(40):
                    ops.conv_gemm(xx, pc, yy, **kw)
                n += 40
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            smi.stop = True
            smi.join()
            us = e0.elapsed_time(e1) * 1e3 / n
            ss = smi.samples[1:-1] if len(smi.samples) > 3 else smi.samples
            pw = sum(a for a, _ in ss) / max(len(ss), 1)
            ck = sum(b for _, b in ss) / max(len(ss), 1)
            lines.append(f"{cin} {k} {dil} {B * L} {name} {us:.1f} {flops / us / 1e6:.1f} {pw:.0f} {ck:.0f}")
            print(lines[-1], flush=True)
        del x, res, y
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
