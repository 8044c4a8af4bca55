#!/usr/bin/env python
"""Secondary benchmark line: KittenTTS tokens -> waveform on one MI355X, synthetic weights of a "nano"-like
shape (mlx_audio_amd/tts/models/kitten_tts/synthetic.py: hidden 128, decoder 256, 24 kHz iSTFTNet x300), canonical T = 80 / F = 264 utterances.

One "step" = one batch of utterances through the whole forward pass.  Two modes in one run: without activation quantisation, and with the module
list the reference's ONNX converter writes (every conv / linear / LSTM input through fake_quant_dynamic_u8: +2 launches per quantised input and
no fused AdaIN / Snake prologues).  Prints ONE JSON line (value = the quantised mode, which is what a converted checkpoint runs); the conv
roofline is measured with events around every conv launch of one instrumented step, like bench.py.
Not the driver's contract line (that is bench.py / Kokoro, config[1]); results are committed under profiles/.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_BF16_PEAK_TFLOPS = 2500.0
T, F = 80, 264


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args(argv)

    from mlx_audio_amd import ops
    from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS
    from mlx_audio_amd.tts.models.kitten_tts.engine import KittenEngine
    from mlx_audio_amd.tts.models.kokoro import synthetic as S

    dev = torch.device("cuda", 0)
    cfg = KS.KITTEN_CONFIG
    w = KS.make_kitten_weights(cfg, seed=0)
    qmods = KS.converter_quant_modules(w)
    B = args.batch
    ids = [S.make_phoneme_ids(T - 2, seed=100 + i).to(dev) for i in range(B)]
    voice = S.make_voice_pack()
    ref = torch.cat([voice[(7 * i) % voice.shape[0]] for i in range(B)], 0).to(dev)
    fds = [S.forced_durations(T, F, seed=i).to(dev) for i in range(B)]

    def measure(quant):
        eng = KittenEngine(w, dict(cfg, activation_quant_modules=qmods if quant else None), device=dev, param_dtype=torch.bfloat16)

        def step():
            return eng.forward(ids, ref, forced_durations=fds)  # SineGen's random inputs are drawn on the device inside the step

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs, _ = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(o.numel() == F * 600 and bool(torch.isfinite(o).all()) for o in outs)
        ops.PROFILE = []
        step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.profile_finalize(ops.PROFILE), None
        flops = sum(p[0] for p in prof)
        ms = sum(p[2].elapsed_time(p[3]) for p in prof)
        return dict(samples_per_s=B * F * 600 * args.steps / dt, ms_per_step=1000 * dt / args.steps, conv_launches=len(prof),
                    conv_ms=ms, conv_tflops=flops / (ms * 1e-3) / 1e12, conv_gflop=flops / 1e9)

    plain = measure(False)
    quant = measure(True)
    res = {
        "metric": "audio samples/sec + real-time factor, KittenTTS (nano-like synthetic shape)", "value": quant["samples_per_s"], "unit": "samples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": quant["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16 weights x fp32 activations (bf16 hi+lo split MFMA, fp32 accumulate); uint8 fake-quantised module inputs",
        "data": "synthetic",
        "config": {"workload": f"KittenTTS tokens->waveform, T={T} F={F} (6.6 s @ 24 kHz), {len(qmods)} fake-quantised modules (converter list)",
                   "utterances_per_gpu": B, "hidden_dim": cfg["hidden_dim"], "max_conv_dim": cfg["max_conv_dim"]},
        "x_realtime": quant["samples_per_s"] / 24000.0,
        "without_activation_quant": {"value": plain["samples_per_s"], "ms_per_step": plain["ms_per_step"], "x_realtime": plain["samples_per_s"] / 24000.0},
        "roofline": {"bound": "mfma", "kernel": "conv_ws4_kernel + conv_gemm_kernel", "achieved": quant["conv_tflops"], "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": quant["conv_tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                     "conv_ms_per_step": quant["conv_ms"], "launches_per_step": quant["conv_launches"], "algorithmic_gflop_per_step": quant["conv_gflop"],
                     "without_activation_quant": {"achieved": plain["conv_tflops"], "conv_ms_per_step": plain["conv_ms"]},
                     "note": "256-channel convs: 128 x 128 tiles of K = 768..2816; the step is launch- and elementwise-bound, the conv share is printed"},
    }
    if not args.no_cpu_baseline:
        from oracle.kitten_ref import KittenRef

        import numpy as np
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        refm = KittenRef(w, dict(cfg, activation_quant_modules=qmods), param_dtype=torch.bfloat16)
        i0, r0, f0 = ids[0].cpu(), ref[0:1].cpu(), fds[0].cpu()
        rng = np.random.default_rng(0)
        rin, nzn = rng.uniform(size=(1, 9)).astype(np.float32), rng.standard_normal((1, 2 * F * 300, 9)).astype(np.float32)
        refm.forward(i0, r0, pred_dur=f0, rand_ini=rin, noise=nzn)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            refm.forward(i0, r0, pred_dur=f0, rand_ini=rin, noise=nzn)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[1]
        res["cpu_baseline"] = {"value": F * 600 / med, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"1 utterance (T={T}, F={F}), 1 warm-up + median of 3; restated reference (oracle/kitten_ref.py, PyTorch-CPU fp32), not MLX"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
