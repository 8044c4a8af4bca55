#!/usr/bin/env python
"""Measurement line for the widened codec rows: Vocos (mel -> waveform, 24 kHz), DAC (44 kHz, 9 codebooks), EnCodec (24 kHz, 6 kbps), SNAC
(24 kHz, 3 code levels) decode and the BigVGAN vocoder (22 kHz, 80 bands) on one MI355X, synthetic weights of the published shapes, inputs resident in HBM before the timed region.

Prints ONE JSON line per codec: value = audio samples decoded per second over the whole batch (and x real time), ms per batch, and a roofline
object for the conv_gemm launches of one instrumented pass (algorithmic FLOPs / summed launch durations against the dense bf16-class MFMA
peak; events on the launch stream, no host synchronisation inside the pass).  Not the driver's contract line.
"""
import argparse
import json
import time

import torch

import _bench_util as U  # noqa: F401  (puts the repo root on sys.path)

MFMA_PEAK_TFLOPS = 2500.0


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = U.ev(), U.ev()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, 1000.0 * (time.perf_counter() - t0) / steps, e0.elapsed_time(e1) / steps


def conv_roofline(fn):
    from mlx_audio_amd import ops

    ops.PROFILE = []
    torch.cuda.synchronize()
    fn()
    prof, ops.PROFILE = ops.profile_finalize(ops.PROFILE), None
    ms = sum(p[2].elapsed_time(p[3]) for p in prof)
    fl, by = sum(p[0] for p in prof), sum(p[1] for p in prof)
    if not prof or ms <= 0:
        return None
    return {"bound": "mfma", "kernel": "conv_gemm (all launches of one decode pass)", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None, "launches": len(prof), "conv_gemm_ms": ms,
            "algorithmic_gflop": fl / 1e9, "hbm_view": {"algorithmic_GB": by / 1e9, "achieved_GBps": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / 8000.0}}


def line(name, workload, sr, n_samples, B, wall_ms, dev_ms, roof, dtype):
    total = n_samples * B
    return {"metric": f"audio samples decoded per second, {name} decode, 1 MI355X", "value": total / (dev_ms * 1e-3), "unit": "samples/s", "n_gpus": 1,
            "ms_per_step": dev_ms, "wall_ms_per_step": wall_ms, "higher_is_better": True, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "batch": B, "samples_per_item": n_samples, "sample_rate": sr}, "x_realtime": total / sr / (dev_ms * 1e-3), "roofline": roof}


def enc_line(name, workload, sr, n_samples, B, wall_ms, dev_ms, roof, dtype, frames, n_codes):
    d = line(name, workload, sr, n_samples, B, wall_ms, dev_ms, roof, dtype)
    d["metric"] = f"audio samples encoded per second, {name} encode, 1 MI355X"
    d["config"].update(frames_per_item=frames, codes_per_item=n_codes)
    if roof is not None:
        roof["kernel"] = "conv_gemm (all launches of one encode pass: encoder convs, in_proj)"
    return d


def encode_lines(args, dev, g, dt16):
    """waveform [B, 1, S] -> codes; synthetic weights of the published shapes (encoder halves from make_*_encoder_weights), audio resident in HBM."""
    B = args.batch

    def audio(sr, hop_multiple):
        n = int(args.seconds * sr) // hop_multiple * hop_multiple
        return (0.3 * torch.randn(B, 1, n, generator=g)).to(dev), n

    if args.only in ("", "dac"):
        from mlx_audio_amd.codec.models.descript import DAC, make_dac_encoder_weights, make_dac_weights

        w = make_dac_weights(1536, [8, 8, 4, 2], 1024, 9, 1024, 8, seed=0)
        w.update(make_dac_encoder_weights(64, [2, 4, 8, 8], 1024, 9, 8, seed=0))
        eng = DAC(encoder_dim=64, encoder_rates=[2, 4, 8, 8], decoder_dim=1536, decoder_rates=[8, 8, 4, 2], n_codebooks=9, codebook_size=1024, codebook_dim=8,
                  sample_rate=44100, weights=w, device=dev)
        x, n = audio(44100, 512)
        fn = lambda: eng.encode(x)[1]  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        print(json.dumps(enc_line("DAC 44 kHz (encoder_dim 64, rates 2/4/8/8, latent 1024, 9 codebooks of 1024 x 8)", f"{B} x {n} samples -> codes (encoder + 9-level residual search)",
                                  44100, n, B, wall, dms, conv_roofline(fn), dt16, int(out.shape[2]), int(out.shape[1] * out.shape[2]))))

    if args.only in ("", "snac"):
        from mlx_audio_amd.codec.models.snac import SNAC, make_snac_encoder_weights, make_snac_weights

        w = make_snac_weights(768, 1024, [8, 8, 4, 2], [4, 2, 1], 4096, 8, True, True, seed=0)
        w.update(make_snac_encoder_weights(48, [2, 4, 8, 8], 768, [4, 2, 1], 8, True, seed=0))
        eng = SNAC(sampling_rate=24000, encoder_dim=48, encoder_rates=[2, 4, 8, 8], decoder_dim=1024, decoder_rates=[8, 8, 4, 2], attn_window_size=None,
                   codebook_size=4096, codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True, weights=w, device=dev)
        x, n = audio(24000, 512 * 4)
        fn = lambda: eng.encode(x)  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        print(json.dumps(enc_line("SNAC 24 kHz (encoder_dim 48, rates 2/4/8/8, depthwise, 3 code levels of 4096 x 8)", f"{B} x {n} samples -> codes (encoder + 3-level multi-scale search)",
                                  24000, n, B, wall, dms, conv_roofline(fn), dt16, int(out[-1].shape[1]), int(sum(c.shape[1] for c in out)))))

    if args.only in ("", "encodec"):
        from mlx_audio_amd.codec.models.encodec import Encodec, make_encodec_encoder_weights, make_encodec_weights

        c = dict(upsampling_ratios=[8, 5, 4, 2], target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0])
        w = make_encodec_weights(c, seed=0)
        w.update(make_encodec_encoder_weights(c, seed=0))
        eng = Encodec(c, weights=w, device=dev)
        x, n = audio(24000, 320)
        xin = x.transpose(1, 2).contiguous()          # [B, samples, 1]
        fn = lambda: eng.encode(xin, None, bandwidth=6.0)[0]  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        T = int(out.shape[-1])
        print(json.dumps(enc_line("EnCodec 24 kHz (32 filters, rates 2/4/5/8, two 512-wide LSTM layers, 6 kbps = 8 of 32 codebooks)",
                                  f"{B} x {n} samples -> codes (SEANet encoder; LSTM = {2 * T} per-step launch pairs; one rvq_encode launch)", 24000, n, B, wall, dms,
                                  conv_roofline(fn), dt16, T, 8 * T)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--encode", action="store_true", help="the ENCODE lines instead (round 5): DAC / SNAC / EnCodec waveform -> codes (encoder + residual codebook search)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    g = torch.Generator().manual_seed(0)
    dt16 = "fp32 checkpoints held as fp16 MFMA images x fp32 activations (fp16 hi+lo split, fp32 accumulate)"

    if args.encode:
        return encode_lines(args, dev, g, dt16)

    if args.only in ("", "vocos"):
        from mlx_audio_amd.codec.models.vocos import Vocos

        cfg = {"feature_extractor": {"class_path": "vocos.feature_extractors.MelSpectrogramFeatures",
                                     "init_args": {"sample_rate": 24000, "n_fft": 1024, "hop_length": 256, "n_mels": 100}},
               "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 512, "intermediate_dim": 1536, "num_layers": 8}},
               "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 512, "n_fft": 1024, "hop_length": 256}}}
        eng = Vocos.from_hparams(cfg, device=dev, seed=0)
        T = int(args.seconds * 24000) // 256
        mel = (torch.randn(B, T, 100, generator=g) * 2.0 - 4.0).to(dev)
        fn = lambda: eng.decode(mel)  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        n = int(out.shape[-1])
        print(json.dumps(line("Vocos (mel-24khz shapes: dim 512, 8 ConvNeXt blocks, iSTFT head n_fft 1024 / hop 256)", f"{B} x {T} mel frames -> waveform", 24000, n, B,
                              wall, dms, conv_roofline(fn), dt16)))

    if args.only in ("", "dac"):
        from mlx_audio_amd.codec.models.descript import DAC

        eng = DAC(encoder_dim=64, encoder_rates=[2, 4, 8, 8], decoder_dim=1536, decoder_rates=[8, 8, 4, 2], n_codebooks=9, codebook_size=1024, codebook_dim=8,
                  sample_rate=44100, device=dev, seed=0)
        T = int(args.seconds * 44100) // 512
        codes = torch.randint(0, 1024, (B, 9, T), generator=g).to(dev)

        def fn():
            z, _, _ = eng.quantizer.from_codes(codes)
            return eng.decode(z)

        out, wall, dms = timed(fn, args.steps, args.warmup)
        n = int(out.shape[1])
        print(json.dumps(line("DAC 44 kHz (decoder_dim 1536, rates 8/8/4/2, 9 codebooks)", f"{B} x {T} code frames -> waveform (from_codes + decoder)", 44100, n, B, wall, dms,
                              conv_roofline(fn), dt16)))

    if args.only in ("", "snac"):
        from mlx_audio_amd.codec.models.snac import SNAC

        eng = SNAC(sampling_rate=24000, encoder_dim=48, encoder_rates=[2, 4, 8, 8], decoder_dim=1024, decoder_rates=[8, 8, 4, 2], attn_window_size=None,
                   codebook_size=4096, codebook_dim=8, vq_strides=[4, 2, 1], noise=True, depthwise=True, device=dev, seed=0)
        T = (int(args.seconds * 24000) // 512) // 4 * 4
        codes = [torch.randint(0, 4096, (B, T // s), generator=g).to(dev) for s in (4, 2, 1)]
        fn = lambda: eng.decode(codes)  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        n = int(out.shape[1])
        print(json.dumps(line("SNAC 24 kHz (decoder_dim 1024, rates 8/8/4/2, depthwise, noise, 3 code levels)", f"{B} x {T} finest-level code frames -> waveform", 24000, n, B,
                              wall, dms, conv_roofline(fn), dt16)))

    if args.only in ("", "encodec"):
        from mlx_audio_amd.codec.models.encodec import Encodec

        eng = Encodec(dict(upsampling_ratios=[8, 5, 4, 2], target_bandwidths=[1.5, 3.0, 6.0, 12.0, 24.0]), device=dev, seed=0)
        T = int(args.seconds * 24000) // 320
        codes = torch.randint(0, 1024, (B, 1, 8, T), generator=g).to(dev)   # 6 kbps: 8 codebooks
        fn = lambda: eng.decode(codes, [None])  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        n = int(out.shape[1])
        print(json.dumps(line("EnCodec 24 kHz (32 filters, rates 8/5/4/2, two 512-wide LSTM layers, 8 of 32 codebooks = 6 kbps)",
                              f"{B} x {T} code frames -> waveform (RVQ decode + SEANet decoder; LSTM = {2 * T} per-step launch pairs)", 24000, n, B, wall, dms,
                              conv_roofline(fn), dt16)))

    if args.only in ("", "bigvgan"):
        from mlx_audio_amd.codec.models.bigvgan import BigVGAN, BigVGANConfig

        cfg = BigVGANConfig(num_mels=80, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], upsample_initial_channel=1536, resblock="1",
                            resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, activation="snakebeta", snake_logscale=True)
        eng = BigVGAN(cfg, device=dev, seed=0)
        T = int(args.seconds * 22050) // 256
        Bb = max(1, B // 4)  # the 22 kHz / 80-band model is ~14 x the FLOPs per sample of the codecs above
        mel = (torch.randn(Bb, 80, T, generator=g) * 0.8).to(dev)
        fn = lambda: eng(mel)  # noqa: E731
        out, wall, dms = timed(fn, args.steps, args.warmup)
        n = int(out.shape[-1])
        print(json.dumps(line("BigVGAN 22 kHz / 80 bands (1536 channels, rates 4/4/2/2/2/2, AMPBlock1 k 3/7/11, SnakeBeta, anti-aliased activations)",
                              f"{Bb} x {T} mel frames -> waveform", 22050, n, Bb, wall, dms, conv_roofline(fn), dt16)))


if __name__ == "__main__":
    main()
