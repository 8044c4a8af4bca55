#!/usr/bin/env python
"""Times the bidirectional LSTM recurrence kernels of csrc/lstm.hip on the shapes of the Kokoro step (H = 256; token-rate L = 80 and frame-rate
L = 264, B utterances): lstm_oct_kernel (eight gate rows x one k-slice per thread) against lstm_kernel (one gate row per thread, MI355_LSTM_OCT=0).

    python tools/bench_lstm.py [--batch 64]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    from mlx_audio_amd import ops

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    print("H L B kernel us_per_launch us_per_step max_abs_diff_vs_old")
    for H, L in ((256, 80), (256, 264), (128, 80), (64, 264)):
        for B in (1, args.batch):
            s = H ** -0.5
            whf = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * s).bfloat16().float()
            whb = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * s).bfloat16().float()
            scaled = ops.pack_lstm_wh_scaled(whf, whb, dev)
            xp = torch.randn(B, L, 8 * H, generator=g).to(dev)
            outs = {}
            for name, env in (("oct", "1"), ("row", "0")):
                os.environ["MI355_LSTM_OCT"] = env
                out = torch.zeros(B, L, 2 * H, device=dev)
                for _ in range(3):
                    ops.lstm_bidir(xp, scaled[0], H, out, wh_f16=True, wh_scale=scaled[1])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.lstm_bidir(xp, scaled[0], H, out, wh_f16=True, wh_scale=scaled[1])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                outs[name] = out
                d = float((outs["oct"] - out).abs().max()) if name == "row" else 0.0
                print(f"{H} {L} {B} {name} {us:.1f} {us / L:.2f} {d:.2e}")
    os.environ["MI355_LSTM_OCT"] = "1"


if __name__ == "__main__":
    main()
