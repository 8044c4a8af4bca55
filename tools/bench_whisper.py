#!/usr/bin/env python
"""Secondary benchmark line: BASELINE config[2] -- Whisper-small STT, 30 s audio -> log-mel + encoder + greedy decode
on one MI355X (synthetic fp16 weights of the exact whisper-small shapes, synthetic audio).

One "step" = one batch of 30 s windows through log-mel (fused STFT/mel kernel), the 12-layer encoder and a fixed number of
greedy decode steps (logit filters on device, EOT not special-cased so every step is timed).  Prints ONE JSON line with
  * value = audio seconds transcribed per wall second (x real time), whole step;
  * encoder / decode split, decode tokens/s;
  * roofline of the decode-step GEMV (HBM: 16-bit weight bytes read per token / time) and of the encoder's attention
    kernel (f32 MFMA: 4*T^2*D flops / time vs the 157 TFLOP/s f32 matrix peak), measured with events on the launch stream;
  * cpu_baseline: the oracle (PyTorch-CPU fp32) on one window's encoder, bounded.
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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64, help="30 s windows per step (64 = 32 minutes of audio per step; <= 8: the one-row / 5..8-row kernels, "
                    "9..64: the rows pipeline)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--decode-steps", type=int, default=64)
    ap.add_argument("--precision", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", choices=["rank0", "none"], default="rank0", help="multi-GPU runs: token sequences back to rank 0, or kept on the rank that made them")
    args = ap.parse_args(argv)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _bench_util as U

    from mlx_audio_amd import dsp, ops
    from mlx_audio_amd.stt.models.whisper import synthetic as WS
    from mlx_audio_amd.stt.models.whisper.engine import WhisperEngine
    from mlx_audio_amd.stt.models.whisper.tokenizer import get_tokenizer

    from mlx_audio_amd import shard

    D_ = U.Dist()   # one process per GPU; the --batch windows are sharded over the ranks (audio out in ONE dense broadcast, token ids back)
    dev = D_.dev
    dims = WS.WHISPER_SMALL
    w = WS.make_whisper_weights(dims, seed=0)
    eng = WhisperEngine(w, dims, device=dev, precision=args.precision)
    tok = get_tokenizer(True, language="en", task="transcribe")
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(0)
    audio = torch.randn((B, 480000), generator=g, device=dev) * 0.1  # resident in HBM (on rank 0: the owner of the requests) before the timed region
    ch = shard.ShardChannel(dev, D_.dist, max_items=max(B, 8), max_tokens=8)
    suppress = [tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech, tok.transcribe, tok.translate, tok.eot]  # EOT suppressed: fixed length

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def step(timers=None):
        e = [ev() for _ in range(4)]
        e[0].record()
        mine, items = shard.scatter_dense(ch, audio if D_.rank == 0 else None, (480000,)) if D_.dist else (audio, list(range(B)))
        if not items:
            for k in (1, 2, 3):
                e[k].record()
            ch.gather([], dtype=torch.int64) if (D_.dist and args.gather == "rank0") else None
            return None
        mel = dsp.log_mel_spectrogram(mine, n_mels=80, padding=480000)[:, :3000]  # whisper.py:_prepare_audio + pad_or_trim
        e[1].record()
        xa = eng.encode(mel.contiguous())
        e[2].record()
        out = eng.decode(None, tok, sample_len=args.decode_steps, suppress_tokens=suppress, audio_features=xa, fixed_steps=True)
        e[3].record()
        if D_.dist and args.gather == "rank0":
            out["gathered"] = ch.gather([t.reshape(-1) for t in out["tokens"]], counts=[int(out["tokens"].shape[1])] * ch.n_items, dtype=torch.int64)
        if timers is not None:
            timers.append(e)
        return out

    for _ in range(args.warmup):
        step()
    D_.fence()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(timers)
    D_.fence()
    dt = D_.max_over_ranks(time.perf_counter() - t0)
    if D_.rank != 0:
        D_.close()
        return None
    assert out["tokens"].shape[1] == 3 + args.decode_steps and bool(torch.isfinite(out["sum_logprobs"]).all())
    assert not (D_.dist and args.gather == "rank0") or (len(out["gathered"]) == B and all(t.numel() == 3 + args.decode_steps for t in out["gathered"]))
    mel_ms = sum(t[0].elapsed_time(t[1]) for t in timers) / args.steps
    enc_ms = sum(t[1].elapsed_time(t[2]) for t in timers) / args.steps
    dec_ms = sum(t[2].elapsed_time(t[3]) for t in timers) / args.steps
    audio_s = 30.0 * B * args.steps
    res = {
        "metric": "audio seconds transcribed per second (x real time), Whisper-small STT, %d MI355X" % D_.world, "value": audio_s / dt, "unit": "x realtime",
        "n_gpus": D_.world, "scaling": "strong", "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000 * dt / args.steps, "higher_is_better": True,
        "dtype": "fp16 weights x fp32 activations (precision %d: %s), f32 MFMA attention" % (
            args.precision, "fp16 hi+lo split" if args.precision == 4 else "single fp16 pass"),
        "data": "synthetic",
        "config": {"workload": "Whisper-small, 30 s windows: log-mel + 12-layer encoder + %d greedy decode steps (filters on device)" % args.decode_steps,
                   "windows_per_step": B, "windows_on_rank0": len(ch.my_items()) if D_.dist else B, "parallelism": f"window-dp{D_.world}", "gather": args.gather,
                   "decode_steps": args.decode_steps},
        "step_runner": "multi-launch native runner (stack_step.cpp)",
        "split_ms": {"logmel": mel_ms, "encoder": enc_ms, "decode": dec_ms},
        "decode_tokens_per_s": B * args.decode_steps / (dec_ms * 1e-3),
        "decode_ms_per_token_step": dec_ms / args.decode_steps,
    }

    # ---- roofline legs: one decode-step GEMV set and one encoder attention, bracketed by events on the launch stream
    nt = dims.n_text_state
    x = torch.randn(min(B, 8), nt, device=dev)
    lin = eng.logits_lin.rm
    y = torch.empty(x.shape[0], lin.n, device=dev)
    for _ in range(3):
        ops.gemv(x, lin, y)
    a0, a1 = ev(), ev()
    a0.record()
    reps = 20
    for _ in range(reps):
        ops.gemv(x, lin, y)
    a1.record()
    torch.cuda.synchronize()
    ms = a0.elapsed_time(a1) / reps
    byts = 2.0 * lin.n * lin.k + 4.0 * x.shape[0] * (lin.k + lin.n)
    res["roofline"] = {"bound": "hbm", "kernel": "gemv_kernel (decode-step logits, N=51865 K=768 fp16 row-major)", "achieved": byts / (ms * 1e-3) / 1e9,
                       "peak": 8000.0, "unit": "GB/s", "frac": byts / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "us_per_launch": ms * 1e3,
                       "algorithmic_bytes_per_launch": byts}
    if B > 8:
        # the dominant kernel of a tall decode step: cross-attention of B windows x heads over 1500 cached keys (K | V in the checkpoint's 16-bit type,
        # head-major): every step streams 2 x B x 1500 x n_state values per layer -- bytes that no batching amortises
        Hh, dh, Tc = dims.n_text_head, nt // dims.n_text_head, dims.n_audio_ctx
        ck = torch.randn(B, Hh, Tc, dh, device=dev).to(eng.kv_dtype)
        cv = torch.randn(B, Hh, Tc, dh, device=dev).to(eng.kv_dtype)
        qx = torch.randn(B, 1, nt, device=dev)
        ax = torch.empty(B, 1, nt, device=dev)
        for _ in range(3):
            ops.flash_attention(qx, ck, cv, ax, heads=Hh, dh=dh, scale=dh ** -0.5, head_major=True)
        c0, c1 = ev(), ev()
        c0.record()
        for _ in range(reps):
            ops.flash_attention(qx, ck, cv, ax, heads=Hh, dh=dh, scale=dh ** -0.5, head_major=True)
        c1.record()
        torch.cuda.synchronize()
        cms = c0.elapsed_time(c1) / reps
        cb = 2.0 * B * Tc * nt * ck.element_size() + 8.0 * B * nt
        res["logits_roofline"] = res["roofline"]
        res["roofline"] = {"bound": "hbm", "kernel": "attn_decode_kernel<64, 16, 16-bit K|V> (decode-step cross-attention: %d windows x %d heads x %d keys)" % (B, Hh, Tc),
                           "achieved": cb / (cms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": cb / (cms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                           "us_per_launch": cms * 1e3, "algorithmic_bytes_per_launch": cb,
                           "share_of_decode_step": 12 * cms / max(dec_ms / args.decode_steps, 1e-9) if dims.n_text_layer == 12 else None}
    T, D, H = dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head
    qkv = torch.randn(B, T, 3 * D, device=dev)
    o = torch.empty(B, T, D, device=dev)
    kv16 = qkv[:, :, D:].to(torch.float16)

    def t_attn(k, v):
        for _ in range(2):
            ops.flash_attention(qkv[:, :, :D], k, v, o, heads=H, dh=D // H)
        a0, a1 = ev(), ev()
        a0.record()
        for _ in range(5):
            ops.flash_attention(qkv[:, :, :D], k, v, o, heads=H, dh=D // H)
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / 5

    ms16, ms32 = t_attn(kv16[:, :, :D], kv16[:, :, D:]), t_attn(qkv[:, :, D:2 * D], qkv[:, :, 2 * D:])
    fl = 4.0 * B * T * T * D
    # dense peak of the 16-bit matrix pipe; Q and P are split hi + lo (2 MFMAs per product), so the algorithmic ceiling of this kernel is 0.5
    res["attention_roofline"] = {"bound": "mfma", "kernel": "flash_attn16_kernel<64, f16> (encoder 1500x1500, fp16 K | V, v_mfma_f32_32x32x16_f16, Q / P hi+lo)",
                                 "achieved": fl / (ms16 * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": fl / (ms16 * 1e-3) / 1e12 / 2500.0,
                                 "ms_per_launch": ms16,
                                 "fp32_kv_kernel": {"kernel": "flash_attn_kernel<64> (f32 MFMA)", "TFLOPs": fl / (ms32 * 1e-3) / 1e12, "ms_per_launch": ms32}}

    # ---- whole-PHASE rooflines: the line's `roofline` is the phase that takes most of the step, priced on SURVEY section 8(d)'s
    # figures, not on its best kernel; the per-kernel legs above stay as sub-fields
    n_here = len(ch.my_items()) if D_.dist else B
    enc_flop = 0.34e12 * n_here                                    # section 8(d): ~0.34 TFLOP per 30 s window (12 x [1500 x 7.08 M + 2 x 1500^2 x 768] MAC + convs)
    ntl, nts, nv = dims.n_text_layer, dims.n_text_state, dims.n_vocab
    w_bytes = 2.0 * (ntl * (6 * nts * nts + 2 * nts * 4 * nts) + nv * nts)                       # 16-bit decoder weights incl. the tied embedding (~0.28 GB)
    ckv_bytes = 2.0 * ntl * n_here * dims.n_audio_ctx * nts * torch.empty(0, dtype=eng.kv_dtype).element_size()   # cross K | V of every window, every step
    step_ms = dec_ms / args.decode_steps
    phases = {
        "encoder": {"bound": "mfma", "achieved": enc_flop / (enc_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": enc_flop / (enc_ms * 1e-3) / 1e12 / 2500.0,
                    "ms": enc_ms, "algorithmic_flop": enc_flop, "what": "whole encoder phase (convs, 12 layers) of the step's windows; section 8(d): 0.34 TFLOP per window"},
        "decode": {"bound": "hbm", "achieved": (w_bytes + ckv_bytes) / (step_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                   "frac": (w_bytes + ckv_bytes) / (step_ms * 1e-3) / 1e9 / 8000.0, "ms_per_token_step": step_ms, "algorithmic_bytes_per_token_step": w_bytes + ckv_bytes,
                   "what": "whole decode token step: 16-bit decoder weights (%.0f MB) + the cross K | V of the step's windows (%.0f MB), every step" % (w_bytes / 1e6, ckv_bytes / 1e6)},
    }
    res["kernel_rooflines"] = {"decode_dominant_kernel": res.pop("roofline"), "logits_gemv": res.pop("logits_roofline", None), "encoder_attention": res.pop("attention_roofline")}
    dom = "decode" if dec_ms >= enc_ms else "encoder"
    res["roofline"] = dict(phases[dom], phase=dom, traffic=None, share_of_step=(dec_ms if dom == "decode" else enc_ms) / max(mel_ms + enc_ms + dec_ms, 1e-9))
    res["phase_rooflines"] = phases

    if not args.no_cpu_baseline:
        from oracle.whisper_ref import WhisperRef

        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 32))
        torch.set_num_threads(cores)
        ref = WhisperRef(w, dims)
        mel1 = WS.make_mel(1, seed=0)
        t1 = time.perf_counter()
        ref.encoder(mel1)
        enc_s = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": 30.0 / enc_s, "unit": "x realtime (encoder only)", "cores": cores, "kind": "port",
                               "sample": "1 window (30 s), encoder only, single run; restated reference (oracle/whisper_ref.py, PyTorch-CPU fp32), not MLX",
                               "host_cores_available": avail}
    print(json.dumps(res))
    D_.close()
    return res


if __name__ == "__main__":
    main()
