#!/usr/bin/env python
"""How long does a reference clip take to become an x-vector?  (Qwen3-TTS voice cloning: once per clip, in front of the first generated frame.)

    python tools/bench_speaker_encoder.py [--out gpurun_out/speaker_encoder.jsonl]

Published widths (``Qwen3TTSSpeakerEncoderConfig()``: 128 mels, 512-wide SE-Res2Net blocks of 8 x 64-channel chunks, 1536-wide MFA / pooling, 1024-d
embedding), seeded parameters, synthetic 24 kHz clips resident in HBM.  One JSON line per (clips, seconds): wall time of ``mel_spectrogram`` + encoder
from events on the launch stream (median of ``--reps`` after warm-up), the conv FLOPs of the encoder per clip, and the restatement
(oracle/ecapa_ref.py, PyTorch-CPU float32) timed on one clip beside it.  Not the driver's contract line (bench.py / Kokoro)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def conv_flops(c, frames):
    ch, ks = c.enc_channels, c.enc_kernel_sizes
    macs = frames * ch[0] * ks[0] * c.mel_dim
    for i in range(1, len(ch) - 1):
        sub = ch[i] // c.enc_res2net_scale
        macs += frames * (ch[i] * ch[i - 1] + (c.enc_res2net_scale - 1) * sub * sub * ks[i] + ch[i] * ch[i]) + 2 * ch[i] * c.enc_se_channels
    macs += frames * (ch[-1] * ch[-1] * ks[-1] + c.enc_attention_channels * 3 * ch[-1] + ch[-1] * c.enc_attention_channels) + c.enc_dim * 2 * ch[-1]
    return 2 * macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from mlx_audio_amd import dsp, ops
    from mlx_audio_amd.tts.models.qwen3_tts import speaker_encoder as SE
    from mlx_audio_amd.tts.models.qwen3_tts.config import Qwen3TTSSpeakerEncoderConfig

    ops.require_gpu()
    dev = "cuda"
    c = Qwen3TTSSpeakerEncoderConfig()
    w = SE.make_speaker_encoder_weights(c, seed=0)
    enc = SE.Qwen3TTSSpeakerEncoder(c, w, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    lines = []
    cpu = None
    if not args.no_cpu_baseline:
        from oracle.ecapa_ref import EcapaRef

        mels = SE.make_mels(1, 281, c.mel_dim, seed=1)
        ref = EcapaRef(w, c)
        torch.set_num_threads(min(16, os.cpu_count() or 1))   # convs this small lose to thread hand-off beyond a few cores (call 49: 14.4 s per clip on 256 threads)
        ref(mels)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 5.0:
            ref(mels)
            n += 1
        cpu = dict(ms_per_clip=(time.perf_counter() - t0) / n * 1e3, cores=torch.get_num_threads(), kind="port", sample="encoder only, one 3 s clip (281 mel frames), float32")
    for clips, seconds in ((1, 3), (1, 10), (1, 30), (16, 10), (64, 10)):
        audio = torch.randn((clips, 24000 * seconds), generator=g, device=dev) * 0.1

        def run():
            mels = dsp.mel_spectrogram(audio, n_fft=1024, num_mels=128, sample_rate=24000, hop_size=256, win_size=1024, fmin=0, fmax=12000)
            return mels, enc(mels)

        mels, emb = run()
        run()
        ts, tm = [], []
        for _ in range(args.reps):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            mels = dsp.mel_spectrogram(audio, n_fft=1024, num_mels=128, sample_rate=24000, hop_size=256, win_size=1024, fmin=0, fmax=12000)
            e1.record()
            enc(mels)
            e2.record()
            torch.cuda.synchronize()
            tm.append(e0.elapsed_time(e1))
            ts.append(e0.elapsed_time(e2))
        ms, ms_mel = sorted(ts)[len(ts) // 2], sorted(tm)[len(tm) // 2]
        frames = mels.shape[1]
        fl = conv_flops(c, frames) * clips
        lines.append(dict(metric="speaker_encoder_ms", clips=clips, seconds_per_clip=seconds, mel_frames=frames, ms=round(ms, 4), ms_mel=round(ms_mel, 4),
                          clips_per_s=round(clips / ms * 1e3, 1), audio_s_per_s=round(clips * seconds / ms * 1e3, 1), conv_gflop=round(fl / 1e9, 3),
                          tflops=round(fl / ms / 1e9, 3), finite=bool(torch.isfinite(emb).all()), dtype="bf16 weights, f32 hi+lo activations", data="synthetic",
                          cpu_baseline=cpu))
    txt = "\n".join(json.dumps(l) for l in lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
