#!/usr/bin/env python
"""Secondary benchmark line: BASELINE config[4] -- CSM-1B (Llama-1B backbone + Llama-100M depth decoder, 32 codebooks x 2051) frame
generation with Mimi RVQ codec decode on one MI355X, synthetic bf16 weights of the exact shapes (sesame.py:204-264, mimi_202407(32)).

Prints ONE JSON line: value = audio seconds per wall second over (prompt prefill + F frames of generate_frame + Mimi decode); split
timings; HBM roofline of a frame (16-bit weight bytes of backbone + 31 depth-decoder steps + heads / frame time).  config[4] asks for
fp8 GEMMs: at 1 row per step these GEMMs are weight-stream bound GEMVs, so ``--weights fp8`` switches every Linear of both stacks and the
heads to OCP e4m3fn weight images with per-row power-of-two scales (half the bytes per step; fp32 FMA on the exactly decoded weights, the
prompt prefill runs the same dequantised values through the bf16 MFMA image); default bf16 = the reference's dtype.  Not the driver's contract line.
"""
import argparse
import json
import time

import torch

import _bench_util as U


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="accepted for symmetry with the other bench lines")
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16")
    ap.add_argument("--nt", default="", help="weight-stream cache policy: comma list of backbone, heads, decoder that load non-temporally (default: the engine's)")
    ap.add_argument("--kv16", action="store_true", help="bf16 KV caches (the reference's cache dtype) instead of float32")
    ap.add_argument("--gather", choices=["rank0", "none"], default="rank0", help="multi-GPU runs: code frames back to rank 0, or kept on the rank that made them")
    args = ap.parse_args(argv)
    fp8 = args.weights == "fp8"

    from mlx_audio_amd.codec.models.mimi import mimi as M
    from mlx_audio_amd.lm.stack import make_lin
    from mlx_audio_amd.tts.models.sesame import engine as E

    from mlx_audio_amd import shard

    D_ = U.Dist()   # one process per GPU; --batch sequences sharded over the ranks
    dev = D_.dev
    cfg = E.csm_1b()
    cfg.text_vocab_size = 4096  # the 128 256-row text table is only gathered from; a small one keeps the setup short
    tiny = E.tiny_csm()
    eng = E.CSMEngine(E.make_csm_weights(tiny, seed=0), tiny, device=dev)
    eng.cfg = cfg
    kvd = torch.bfloat16 if args.kv16 else torch.float32
    eng.backbone = U.build_deep_stack(cfg.backbone, dev, seed=1, weight_format=args.weights, kv_dtype=kvd)
    eng.decoder = U.build_deep_stack(cfg.decoder, dev, seed=2, weight_format=args.weights, kv_dtype=kvd)
    if args.nt:
        nts = set(args.nt.split(","))
        eng.set_stream_policy(backbone=int("backbone" in nts), heads=int("heads" in nts), decoder=int("decoder" in nts))
    eng.backbone_cache = eng.backbone.make_cache()
    eng.decoder_cache = eng.decoder.make_cache()
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(torch.bfloat16).to(torch.float32)

    D, Dd, V, nb = cfg.backbone.d_model, cfg.decoder.d_model, cfg.audio_vocab_size, cfg.audio_num_codebooks
    eng.projection = make_lin(rnd(Dd, D, std=1.0 / D ** 0.5), None, dev, fp8=fp8)
    eng.c0_head = make_lin(rnd(V, D, std=4.0 / D ** 0.5), None, dev, fp8=fp8)
    eng.heads = [make_lin(rnd(V, Dd, std=4.0 / Dd ** 0.5), None, dev, fp8=fp8) for _ in range(nb - 1)]
    eng.table = torch.cat([rnd(V * nb, D, std=0.5), rnd(cfg.text_vocab_size, D, std=0.5)], 0).contiguous().to(dev)
    eng.slot_offs = torch.tensor([i * V for i in range(nb)] + [nb * V], dtype=torch.int32, device=dev)
    mcfg = M.mimi_202407(32)
    mimi = M.MimiDecoder(M.make_mimi_decoder_weights(mcfg, seed=0), mcfg, device=dev)

    B, F, S = args.batch, args.frames, args.prompt
    requests = [torch.randint(0, cfg.text_vocab_size, (S,), generator=g) for _ in range(B)]   # text token ids, owned by rank 0
    ch = shard.ShardChannel(dev, D_.dist, max_items=max(B, 8), max_tokens=max(S, 8))
    last = {}

    def run_local(items, ids):
        if not items:
            last.update(fr=None, wav=None)
            return []
        b = len(items)
        toks = torch.zeros(b, S, nb + 1, dtype=torch.long, device=dev)
        mask = torch.zeros(b, S, nb + 1, dtype=torch.bool, device=dev)
        toks[:, :, -1] = torch.stack([i.long() for i in ids])
        mask[:, :, -1] = True
        out = eng.generate(toks, mask, F, temperature=0.0, poll=10 ** 9)
        last["ev"].record()
        fr = out["frames"]
        codes = (fr % mcfg.quantizer_bins).permute(0, 2, 1).contiguous()
        last.update(fr=fr, wav=mimi(codes))
        return [f.reshape(-1) for f in fr]

    def step(timers=None):
        e = [U.ev() for _ in range(3)]
        last["ev"] = e[1]
        e[0].record()
        got = shard.sharded_decode(ch, requests if D_.rank == 0 else None, run_local, dtype=torch.int64, gather=args.gather)
        e[2].record()
        if timers is not None:
            timers.append(e)
        return last.get("fr"), last.get("wav")

    for _ in range(args.warmup):
        step()
    D_.fence()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fr, wav = step(timers)
    D_.fence()
    dt = D_.max_over_ranks(time.perf_counter() - t0)
    if D_.rank != 0:
        D_.close()
        return None
    n = fr.shape[1]
    assert n >= 1 and wav.shape[-1] == n * 1920 and bool(torch.isfinite(wav).all())
    lm_ms = sum(t[0].elapsed_time(t[1]) for t in timers) / args.steps
    dec_ms = sum(t[1].elapsed_time(t[2]) for t in timers) / args.steps
    frame_ms = lm_ms / n
    bpw = 1.0 if fp8 else 2.0
    wbytes = U.stack_weight_bytes(cfg.backbone, bpw) + (nb - 1) * U.stack_weight_bytes(cfg.decoder, bpw) + bpw * (V * D + (nb - 1) * (V * Dd + Dd * D))
    res = {
        "metric": "audio seconds generated per second (x real time), CSM-1B generate_frame + Mimi decode, %d MI355X" % D_.world, "value": B * n * 0.08 * args.steps / dt,
        "unit": "x realtime", "n_gpus": D_.world, "scaling": "strong", "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000 * dt / args.steps, "higher_is_better": True,
        "dtype": ("fp8 e4m3fn weights (per-row 2^k scales) x fp32 activations (GEMV fp32 FMA on exactly decoded weights)" if fp8 else
                  "bf16 weights x fp32 activations (GEMV fp32 FMA on bf16 weights)") + ("; 5..8 sequences: v_mfma_f32_16x16x32 on the weights' own type" if B >= 5 else ""),
        "data": "synthetic",
        "config": {"workload": "CSM-1B: prompt %d tokens, %d frames x (backbone step + 31 depth-decoder steps, sampling on device), Mimi decode (32 codebooks)" % (S, n),
                   "sequences": B, "sequences_on_rank0": len(ch.my_items()), "parallelism": f"sequence-dp{D_.world}", "gather": args.gather, "frames": n, "temperature": 0.0, "weights": args.weights, "nt": args.nt or "engine default", "kv_cache": "bf16" if args.kv16 else "fp32"},
        "step_runner": U.step_runner(), "split_ms": {"frame_loop": lm_ms, "mimi_decode": dec_ms}, "ms_per_frame": frame_ms, "frames_per_s": B * n / (lm_ms * 1e-3),
        "mimi_samples_per_s": B * n * 1920 / (dec_ms * 1e-3),
        "roofline": {"bound": "hbm", "kernel": "gemv_kernel (all decode-step Linear layers of one frame)", "achieved": wbytes / (frame_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": wbytes / (frame_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "algorithmic_bytes_per_frame": wbytes,
                     "note": "whole-frame figure: includes attention, norms, sampling and launch gaps"},
    }
    if B == 1 and D_.world == 1:
        # time to first audio of a streamed utterance (sesame.py:825-860; the reference's own published metric is TTFB): prompt prefill +
        # `k` frames + the Mimi streaming decoder's first chunk, wall clock to the synchronised waveform
        from mlx_audio_amd.codec.models.mimi.mimi import MimiStreamingDecoder

        toks = torch.zeros(1, S, nb + 1, dtype=torch.long, device=dev)
        mask = torch.zeros(1, S, nb + 1, dtype=torch.bool, device=dev)
        toks[:, :, -1] = requests[0].long().to(dev)
        mask[:, :, -1] = True
        ttfb = {}
        for k in (1, 6):   # one frame (80 ms of audio), and the reference's default streaming_interval 0.5 s = 6 frames
            best = None
            for _ in range(3):
                sd = MimiStreamingDecoder(mimi)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                it = eng.generate_chunks(toks, mask, max(F, k), chunk=k, temperature=0.0)
                blk = next(it)
                wav1 = sd.decode_frames((blk[0] % mcfg.quantizer_bins).t()[None].contiguous())
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                it.close()
                assert wav1.shape[-1] == k * 1920 and eng.frames_generated == k
                best = dt1 if best is None else min(best, dt1)
            ttfb["%d_frame%s" % (k, "" if k == 1 else "s")] = 1000.0 * best
        res["ttfb_ms"] = ttfb
        res["ttfb_note"] = "prompt prefill (%d tokens) + k frames + Mimi streaming decode of those frames, best of 3; audio leaves while the frame loop runs" % S
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = U.cpu_frame_baseline([(cfg.backbone, 1), (cfg.decoder, nb - 1)], B, context=S)
    print(json.dumps(res))
    D_.close()
    return res


if __name__ == "__main__":
    main()
