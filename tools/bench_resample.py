"""Times ``mi355_resample_poly`` on the GPU: R clips of S seconds, ``orig`` -> ``target`` Hz, inputs resident in HBM.  One JSON line:
input samples / s, the kernel's algorithmic bytes (4 B read per input sample + 4 B written per output sample) over its measured time against the
8 TB/s HBM peak, and its float64 multiply-adds against the time.  The host path of the reference (scipy, one core) is timed on one clip beside it."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--orig", type=int, default=44100)
    ap.add_argument("--target", type=int, default=16000)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from mlx_audio_amd.resample import polyphase_table, resample_audio_array, resample_on_device

    n = int(a.seconds * a.orig)
    x = torch.randn(a.rows, n, device="cuda")
    for _ in range(3):
        y = resample_on_device(x, a.orig, a.target)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        y = resample_on_device(x, a.orig, a.target)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    up, down, table, first, n_out = polyphase_table(a.orig, a.target, n)
    byts = 4.0 * a.rows * (n + n_out)
    fma = float(a.rows) * n_out * table.shape[0]
    h = x[0].cpu().numpy()
    t0 = time.perf_counter()
    ref = resample_audio_array(h, a.orig, a.target)
    host_s = time.perf_counter() - t0
    err = float(np.abs(y[0].cpu().numpy() - ref).max())
    print(json.dumps({"metric": "resample_input_samples_per_s", "value": a.rows * n / (ms * 1e-3), "ms": ms, "rows": a.rows, "n_in": n, "n_out": n_out,
                      "up": up, "down": down, "taps_per_output": int(table.shape[0]), "hbm_frac": byts / (ms * 1e-3) / 8e12,
                      "f64_tflops": 2 * fma / (ms * 1e-3) / 1e12, "host_scipy_input_samples_per_s": n / host_s, "max_abs_vs_host": err}))


if __name__ == "__main__":
    main()
