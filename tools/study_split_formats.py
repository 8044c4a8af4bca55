#!/usr/bin/env python
"""What would a cheaper LO pass cost in accuracy?  (CPU study, tools only -- not product, not a test.)

The vocoder convs multiply float32 activations by 16-bit weights as TWO matrix passes: hi = bf16(x) and lo = bf16(x - hi) (``precision 2``).
Call 53 measured the 8-bit matrix shapes of this part at 2.0-2.2 x the bf16 rate under the power limit, so the lo pass -- 2^-8 of the magnitude, a few
significant bits needed -- could run there at ~0.46 of a pass.  This script answers the accuracy side on the CPU, with the Kokoro restatement
(oracle/kokoro_ref.py) at the published widths: the decoder (Decoder + Generator, all 72 conv layers) is run once in float32 and once per SCHEME with every
conv replaced by the arithmetic the kernel would perform:

    y = conv(hi(x), w_hi) + conv(lo(x - hi(x)), w_lo)          accumulation in float32, bias in float32

    hi formats   bf16 | fp16                         (round to nearest even)
    lo formats   none | bf16 | fp16 | e4m3 with an E8M0 scale per 32 consecutive input channels (OCP MX: shared exponent = floor(log2(amax)) - 8,
                 elements saturate at 448) -- what ``v_mfma_scale_f32_32x32x64_f8f6f4`` consumes
    w_lo         the 16-bit weights, or their MX e4m3 image (scale per 32 input channels of one (output channel, tap)) when lo is e4m3

and the waveform is compared with the float32 run: max-abs / peak and SNR -- the two quantities the device tests bound (2e-3 and 50 dB).

    python tools/study_split_formats.py [--frames 66] [--out profiles/r3_split_format_study.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rnd(x, fmt):
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if fmt == "fp16":
        return x.to(torch.float16).to(torch.float32)
    raise KeyError(fmt)


def mx_e4m3(x, dim):
    """OCP MX e4m3 image of ``x`` with one power-of-two scale per 32 consecutive elements along ``dim`` (the K axis of the product)."""
    x = x.movedim(dim, -1)
    shp = x.shape
    C = shp[-1]
    pad = (-C) % 32
    xp = F.pad(x, (0, pad)).reshape(*shp[:-1], -1, 32)
    amax = xp.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -126))) - 8.0
    scale = torch.exp2(e.clamp(-127.0, 127.0))
    q = (xp / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale
    q = torch.where(amax > 0, q, torch.zeros_like(q))
    return q.reshape(*shp[:-1], -1)[..., :C].movedim(-1, dim)


def mx_small(x, dim, ebits, mbits, bump=False):
    """OCP MX fp6 / fp4 image (e2m3, e3m2, e2m1): one power-of-two scale per 32 elements along ``dim``; shared exponent = floor(log2(amax)) - emax_elem,
    elements round to nearest even and saturate at the format's maximum."""
    bias = (1 << (ebits - 1)) - 1
    emax = (1 << ebits) - 1 - bias            # no inf / nan encodings in fp6 / fp4
    fmax = (2.0 - 2.0 ** -mbits) * 2.0 ** emax
    x = x.movedim(dim, -1)
    shp = x.shape
    C = shp[-1]
    pad = (-C) % 32
    xp = F.pad(x, (0, pad)).reshape(*shp[:-1], -1, 32).double()
    amax = xp.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -126))) - emax
    if bump:   # no saturation: one more exponent step whenever amax / 2^e would exceed the format's maximum
        e = torch.where(amax / torch.exp2(e) > fmax, e + 1.0, e)
    scale = torch.exp2(e.clamp(-127.0, 127.0))
    v = (xp / scale).clamp(-fmax, fmax)
    ex = torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -40))).clamp_min(1 - bias)   # subnormals share the smallest normal exponent
    step = torch.exp2(ex - mbits)
    q = (torch.round(v / step) * step).clamp(-fmax, fmax) * scale     # torch.round = half to even
    q = torch.where(amax > 0, q, torch.zeros_like(q)).float()
    return q.reshape(*shp[:-1], -1)[..., :C].movedim(-1, dim)


SMALL = {"e2m3": (2, 3), "e3m2": (3, 2), "e2m1": (2, 1), "e2m1_bump": (2, 1, True), "e2m1_bump_act": (2, 1, True), "e2m1_bump_w": (2, 1, False)}


class Scheme:
    def __init__(self, name, hi, lo):
        self.name, self.hi, self.lo = name, hi, lo

    def split(self, x):
        """x [B, C, L] float32 -> (hi, lo or None)"""
        h = rnd(x, self.hi)
        if self.lo == "none":
            return h, None
        r = x - h
        if self.lo in ("bf16", "fp16"):
            return h, rnd(r, self.lo)
        if self.lo in SMALL:
            return h, mx_small(r, 1, *SMALL[self.lo])
        return h, mx_e4m3(r, 1)

    def w_lo(self, w):
        """w (C_out, K, C_in / groups): the lo pass's weights"""
        if self.lo == "e2m1_bump_act":
            return mx_small(w, 2, 2, 1, False)
        if self.lo == "e2m1_bump_w":
            return mx_small(w, 2, 2, 1, True)
        if self.lo in SMALL:
            return mx_small(w, 2, *SMALL[self.lo])
        return mx_e4m3(w, 2) if self.lo == "e4m3" else w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=66, help="predicted frames F of the utterance (600 samples each); 264 = the canonical short sentence")
    ap.add_argument("--tokens", type=int, default=22)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from oracle import kokoro_ref as KR

    torch.manual_seed(0)
    w = S.make_kokoro_weights()
    ref = KR.KokoroRef(w, S.KOKORO_CONFIG)
    ids = S.make_phoneme_ids(args.tokens - 2)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(len(ids), args.frames)
    rng = np.random.default_rng(1234)
    up = 600
    ri = rng.uniform(size=(1, 9)).astype(np.float32)
    nz = rng.standard_normal((1, args.frames * up, 9)).astype(np.float32)
    t0 = time.time()
    audio_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
    t_plain = time.time() - t0
    audio_ref = audio_ref.reshape(-1)
    peak = float(audio_ref.abs().max())
    p = ref.p.sub("decoder")
    cfg = S.KOKORO_CONFIG["istftnet"]
    s = ref_s[:, :128].to(torch.float32)
    plain1, plainT = KR.conv1d_mlx, KR.conv_transpose1d_mlx

    def run(sch):
        def c1(x, wt, b, stride=1, padding=0, dilation=1, groups=1):
            if x.dtype != torch.float32:
                return plain1(x, wt, b, stride, padding, dilation, groups)
            h, l = sch.split(x)
            wt32 = wt.to(torch.float32)
            y = plain1(h, wt32, None, stride, padding, dilation, groups)
            if l is not None:
                y = y + plain1(l, sch.w_lo(wt32), None, stride, padding, dilation, groups)
            return y if b is None else y + b.to(torch.float32).view(1, -1, 1)

        def ct(x, wt, b, stride=1, padding=0, groups=1):
            h, l = sch.split(x)
            wt32 = wt.to(torch.float32)
            y = plainT(h, wt32, None, stride, padding, groups)
            if l is not None:
                # K axis of a transposed conv's product = input channels = the LAST axis of the MLX weight for groups == 1 (first for depthwise: one channel)
                y = y + plainT(l, sch.w_lo(wt32) if groups == 1 else wt32, None, stride, padding, groups)
            return y if b is None else y + b.to(torch.float32).view(1, -1, 1)

        KR.conv1d_mlx, KR.conv_transpose1d_mlx = c1, ct
        try:
            with torch.no_grad():
                a = KR.decoder(p, tr["asr"], tr["f0"], tr["n"], s, cfg, ri, nz)[0].reshape(-1)
        finally:
            KR.conv1d_mlx, KR.conv_transpose1d_mlx = plain1, plainT
        return a

    # the decoder alone, unchanged arithmetic, must reproduce the full forward (the harness check)
    with torch.no_grad():
        chk = KR.decoder(p, tr["asr"], tr["f0"], tr["n"], s, cfg, ri, nz)[0].reshape(-1)
    assert float((chk - audio_ref).abs().max()) == 0.0
    schemes = [Scheme("bf16 hi only            (precision 1, 1 pass)", "bf16", "none"),
               Scheme("fp16 hi only            (precision 3, 1 pass)", "fp16", "none"),
               Scheme("bf16 hi + bf16 lo       (precision 2, 2 passes: today)", "bf16", "bf16"),
               Scheme("fp16 hi + fp16 lo       (precision 4, 2 passes)", "fp16", "fp16"),
               Scheme("bf16 hi + MX e4m3 lo    (1 + ~0.46 passes)", "bf16", "e4m3"),
               Scheme("fp16 hi + MX e4m3 lo    (1 + ~0.46 passes)", "fp16", "e4m3"),
               Scheme("fp16 hi + MX fp6 e2m3 lo (1 + ~0.27 passes)", "fp16", "e2m3"),
               Scheme("fp16 hi + MX fp6 e3m2 lo (1 + ~0.27 passes)", "fp16", "e3m2"),
               Scheme("fp16 hi + MX fp4 e2m1 lo (1 + ~0.27 passes)", "fp16", "e2m1"),
               Scheme("fp16 hi + MX fp4 e2m1 lo, no-saturation scales (both)", "fp16", "e2m1_bump"),
               Scheme("fp16 hi + MX fp4 e2m1 lo, no-saturation scales (activations only)", "fp16", "e2m1_bump_act"),
               Scheme("fp16 hi + MX fp4 e2m1 lo, no-saturation scales (weights only)", "fp16", "e2m1_bump_w")]
    lines = [f"Kokoro-82M decoder (published widths, seeded parameters), T = {len(ids)} tokens, F = {args.frames} frames = {audio_ref.numel()} samples, peak {peak:.3f}; "
             f"float32 restatement {t_plain:.1f} s on {torch.get_num_threads()} threads",
             "scheme | max-abs / peak | SNR dB | device bars: 2e-3 and 50 dB"]
    for sch in schemes:
        t0 = time.time()
        a = run(sch)
        err = float((a - audio_ref).abs().max()) / peak
        snr = float(10 * torch.log10((audio_ref.double() ** 2).sum() / ((a.double() - audio_ref.double()) ** 2).sum().clamp_min(1e-300)))
        ok = "ok" if err <= 2e-3 and snr >= 50 else "FAILS"
        lines.append(f"{sch.name} | {err:.2e} | {snr:6.1f} | {ok}   ({time.time() - t0:.0f} s)")
        print(lines[-1], flush=True)
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
