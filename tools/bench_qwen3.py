#!/usr/bin/env python
"""Secondary benchmark line: BASELINE config[3] -- Qwen3-TTS-1.7B talker + code predictor frame loop and codec decode, the whole 64-utterance
batch on one GPU (``--batch 8`` = the per-GPU share when the batch is sharded over 8 GPUs), synthetic bf16 weights of the 1.7B shapes (talker hidden 2048 / inter 6144 / 28 L / 16-8 heads,
code predictor config.py:36-52, codec decoder config.py:110-136), temperature 0, on one MI355X.

Prints ONE JSON line: value = audio seconds generated per wall second over (prefill + F frames + codec decode of the F frames);
split timings; roofline of the decode step against HBM: 16-bit weight bytes every frame must stream (talker + 15 code-predictor steps +
heads) / measured frame time.  Not the driver's contract line (bench.py / Kokoro); committed under profiles/.
"""
import argparse
import json
import time

import torch

import _bench_util as U


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--prompt", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", choices=["rank0", "none"], default="rank0",
                    help="multi-GPU runs: the code sequences come back to rank 0 in one exact-size all_to_all (rank0) or stay on the rank that made them (none)")
    args = ap.parse_args(argv)

    from mlx_audio_amd import ops
    from mlx_audio_amd.lm.stack import make_lin
    from mlx_audio_amd.tts.models.qwen3_tts import synthetic as QS
    from mlx_audio_amd.tts.models.qwen3_tts import talker as T
    from mlx_audio_amd.tts.models.qwen3_tts.codec import Qwen3CodecDecoder
    from mlx_audio_amd.tts.models.qwen3_tts.config import Qwen3TTSTokenizerDecoderConfig, talker_1p7b

    from mlx_audio_amd import shard

    D = U.Dist()   # one process per GPU; the batch of --batch utterances is sharded over the ranks (strong scaling: BASELINE config[3] = 64 over 8)
    dev = D.dev
    cfg = talker_1p7b()
    cp = cfg.code_predictor_config
    # --- build the engine from tiny parameters, then swap in full-size stacks / heads (see _bench_util.build_deep_stack)
    tiny = T.tiny_talker_config()
    eng = T.Qwen3Talker(T.make_talker_weights(tiny, seed=0), tiny, device=dev)
    eng.cfg = cfg
    eng.talker = U.build_deep_stack(T.talker_stack_config(cfg), dev, seed=1)
    eng.cp = U.build_deep_stack(T.talker_stack_config(cp), dev, seed=2)
    g = torch.Generator().manual_seed(0)

    def rnd(n, k, std):
        return (torch.randn(n, k, generator=g) * std).to(torch.bfloat16).to(torch.float32)

    H = cfg.hidden_size
    eng.codec_head = make_lin(rnd(cfg.vocab_size, H, 4.0 / H ** 0.5), None, dev)
    eng.lm_heads = [make_lin(rnd(cp.vocab_size, cp.hidden_size, 4.0 / cp.hidden_size ** 0.5), None, dev) for _ in range(cfg.num_code_groups - 1)]
    eng.mtp = make_lin(rnd(cp.hidden_size, H, 1.0 / H ** 0.5), torch.zeros(cp.hidden_size), dev)
    tabs = [rnd(cfg.vocab_size, H, 0.5)] + [rnd(cp.vocab_size, H, 0.5) for _ in range(cfg.num_code_groups - 1)]
    offs, r = [], 0
    for t in tabs:
        offs.append(r)
        r += t.shape[0]
    eng.codec_table = torch.cat(tabs, 0).contiguous().to(dev)
    eng.codec_offs = torch.tensor(offs, dtype=torch.int32, device=dev)
    sup = torch.zeros(cfg.vocab_size)
    sup[[i for i in range(cfg.vocab_size - 1024, cfg.vocab_size)]] = -float("inf")  # EOS suppressed too: fixed number of frames
    eng.suppress_mask = sup.to(dev)
    ccfg = Qwen3TTSTokenizerDecoderConfig()
    codec = Qwen3CodecDecoder(QS.make_codec_decoder_weights(ccfg, seed=0), ccfg, device=dev)

    B, F = args.batch, args.frames
    # requests = token ids owned by rank 0; a request's prompt embeddings are a lookup into a table every rank holds (what text_embedding +
    # text_projection are in the model), so only the ids cross ranks (ShardChannel.scatter_requests: one broadcast)
    table = (torch.randn(512, H, generator=g) * 0.5).to(dev)
    requests = [torch.randint(0, 512, (args.prompt,), generator=g) for _ in range(B)]   # drawn on every rank (same generator state), passed by rank 0 only
    trail_all = (torch.randn(B, 16, H, generator=g) * 0.5).to(dev)
    pad = (torch.randn(1, 1, H, generator=g) * 0.5).to(dev)
    ch = shard.ShardChannel(dev, D.dist, max_items=max(B, 8), max_tokens=max(args.prompt, 8))
    last = {}

    def run_local(items, ids):
        if not items:
            last.update(codes=None, wav=None)
            return []
        pre = table[torch.stack([i.long() for i in ids])]
        out = eng.generate(pre, trail_all[torch.tensor(items, device=dev)], pad, F, temperature=0.0, poll=10 ** 9)
        last["ev"].record()
        codes = (out["codes"] % ccfg.codebook_size).permute(0, 2, 1).contiguous()  # [b, 16, F]
        last.update(codes=out["codes"], wav=codec.chunked_decode(codes, chunk_size=300, left_context_size=25))
        return [c.reshape(-1) for c in out["codes"]]

    def step(timers=None):
        e = [U.ev() for _ in range(3)]
        last["ev"] = e[1]
        e[0].record()
        got = shard.sharded_decode(ch, requests if D.rank == 0 else None, run_local, dtype=torch.int64, gather=args.gather)
        e[2].record()
        if timers is not None:
            timers.append(e)
        return got, last.get("wav")

    for _ in range(args.warmup):
        step()
    D.fence()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got, wav = step(timers)
    D.fence()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    n_local = len(ch.my_items())
    if n_local:
        assert last["codes"].shape[1] == F and wav.shape[-1] == F * 1920 and bool(torch.isfinite(wav).all())
    if D.rank == 0 and args.gather == "rank0":
        assert len(got) == B and all(c.numel() == F * cfg.num_code_groups for c in got)
    if D.rank != 0:
        D.close()
        return None
    B_local = n_local
    lm_ms = sum(t[0].elapsed_time(t[1]) for t in timers) / args.steps     # rank 0's share: request broadcast + frame loop
    dec_ms = sum(t[1].elapsed_time(t[2]) for t in timers) / args.steps    # ... codec decode + the gather of the code sequences
    audio_s = B * F * 0.08 * args.steps
    frame_ms = lm_ms / F  # includes the (short) prefill
    wbytes = U.stack_weight_bytes(eng.talker.cfg) + (cfg.num_code_groups - 1) * U.stack_weight_bytes(eng.cp.cfg) + 2.0 * (
        cfg.vocab_size * H + (cfg.num_code_groups - 1) * (cp.vocab_size * cp.hidden_size + cp.hidden_size * H))
    res = {
        "metric": "audio seconds generated per second (x real time), Qwen3-TTS-1.7B talker + code predictor + codec decode, %d MI355X" % D.world,
        "value": audio_s / dt, "unit": "x realtime", "n_gpus": D.world, "scaling": "strong", "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000 * dt / args.steps,
        "higher_is_better": True, "dtype": "bf16 weights x fp32 activations (decode steps: fp32 FMA GEMV on bf16 weights at <= 4 rows, bf16 hi+lo MFMA above -- rows pipeline at 9..64 rows; bf16 hi+lo MFMA in prefill and codec)", "data": "synthetic",
        "config": {"workload": "Qwen3-TTS-1.7B: prefill %d + %d frames x (talker step + 15 code-predictor steps + sampling on device), then codec decode" % (args.prompt, F),
                   "utterances": B, "utterances_on_rank0": B_local, "parallelism": f"utterance-dp{D.world}", "gather": args.gather, "frames": F, "temperature": 0.0},
        "step_runner": U.step_runner(), "split_ms": {"frame_loop": lm_ms, "codec_decode": dec_ms}, "ms_per_frame": frame_ms, "frames_per_s": B_local * F / (lm_ms * 1e-3),
        "codec_samples_per_s": B_local * F * 1920 / (dec_ms * 1e-3),
        "collectives_per_step": ch.collectives // max(1, args.steps + args.warmup),
        "roofline": {"bound": "hbm", "kernel": ("rows_gemm_kernel + rows_finish_kernel" if B_local > 8 else "gemv kernels") + " (all decode-step Linear layers of one frame)", "achieved": wbytes / (frame_ms * 1e-3) / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": wbytes / (frame_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_frame": wbytes,
                     "note": "whole-frame figure (weights streamed once per frame / wall time of a frame): includes attention, norms, sampling and launch gaps"},
    }
    if D.world == 1:
        # time to first audio of ONE streamed utterance (qwen3_tts.py:1426-1465; the reference's published metric: 84.8 ms TTFB at batch 1 on Apple silicon,
        # 6-bit model): prefill + k frames (talker step + 15 code-predictor steps each) + decoder.streaming_step of those frames
        pre1 = table[requests[0].long()[None].to(dev)]
        ttfb = {}
        for k in (1, 25):   # one frame (80 ms of audio), and the reference's default streaming_interval 2.0 s = 25 frames
            best = None
            for _ in range(3):
                st = codec.new_stream(1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                it = eng.generate_iter(pre1, trail_all[:1], pad, max(F, k), temperature=0.0, chunk=k)
                blk = next(it)["block"]
                wav1 = codec.streaming_step((blk % ccfg.codebook_size).permute(0, 2, 1).contiguous(), st)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                it.close()
                assert wav1.shape[-1] == k * 1920 and eng.frames_generated == k
                best = dt1 if best is None else min(best, dt1)
            ttfb["%d_frame%s" % (k, "" if k == 1 else "s")] = 1000.0 * best
        res["ttfb_ms"] = ttfb
        res["ttfb_note"] = "one utterance: prefill (%d positions) + k frames + codec streaming_step of those frames, best of 3; audio leaves while the frame loop runs" % args.prompt
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = U.cpu_frame_baseline([(eng.talker.cfg, 1), (eng.cp.cfg, cfg.num_code_groups - 1)], B, context=args.prompt)
    print(json.dumps(res))
    D.close()
    return res


if __name__ == "__main__":
    main()
