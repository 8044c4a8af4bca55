#!/usr/bin/env python
"""Diagnostic: B IDENTICAL canonical utterances (same ids, style, forced durations, SineGen inputs) through one engine call.  Every row of every
intermediate tensor must then equal row 0 (each row walks the same tiles of the same kernels), so any spread over the rows is a defect of the
batched launch itself -- no oracle needed.  Prints, per traced tensor in forward order, the worst row's max |row - row 0| / max |row 0|.

    python tools/diag_batch_rows.py --precision 5 --batch 64 [--rows]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", type=int, default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rows", action="store_true", help="also list the rows that differ from row 0 in the final waveform")
    ap.add_argument("--repeat", type=int, default=1, help="run the same call this many times and compare the waveforms of consecutive runs (run-to-run determinism)")
    args = ap.parse_args()
    from mlx_audio_amd.tts.models.kokoro import synthetic as S
    from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine

    eng = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG, precision=args.precision)
    B = args.batch
    ids = S.make_phoneme_ids(78)
    ref_s = S.make_voice_pack()[len(ids) - 3]
    fd = S.forced_durations(80, 264)
    rng = np.random.default_rng(1234)
    ri = torch.from_numpy(rng.uniform(size=(1, 9)).astype(np.float32)).cuda().expand(B, -1).contiguous()
    nz = torch.from_numpy(rng.standard_normal((1, 2 * 264 * 300, 9)).astype(np.float32)).cuda().expand(B, -1, -1).contiguous()
    prev = None
    for rep in range(args.repeat):
        outs, _, tr = eng.forward([ids] * B, ref_s.repeat(B, 1), forced_durations=[fd] * B, rand_ini=ri, noise=nz, return_intermediates=True)
        torch.cuda.synchronize()
        wav = torch.stack(outs)
        if prev is not None:
            print(f"run {rep} vs run {rep - 1}: max |diff| = {float((wav - prev).abs().max()):.3e}")
        prev = wav.clone()
    order = ["d", "t_en", "en", "asr", "f0", "n", "dec_in", "enc", "dec0", "dec1", "dec2", "xg", "har_src", "har", "nconv0", "nres0", "xu0", "stage0",
             "nconv1", "nres1", "xu1", "stage1", "post"]
    print(f"precision {eng.precision}, {B} identical utterances: worst row spread per tensor (max |row - row 0| / max |row 0|)")
    for k in order:
        v = tr.get(k)
        if v is None or v.shape[0] != B:
            continue
        v = v.reshape(B, -1).float()
        ref = v[0:1]
        dev = (v - ref).abs().amax(dim=1)
        scale = float(ref.abs().max()) + 1e-30
        w = int(dev.argmax())
        nbad = int((dev > 0).sum())
        print(f"  {k:8s} shape {tuple(tr[k].shape)}  worst {float(dev[w]) / scale:.3e} (row {w})  rows != row 0: {nbad}")
    dev = (wav - wav[0:1]).abs().amax(dim=1)
    peak = float(wav[0].abs().max())
    print(f"  waveform peak {peak:.3f}: worst row spread {float(dev.max()) / peak:.3e} (row {int(dev.argmax())}), rows != row 0: {int((dev > 0).sum())}")
    if args.rows:
        print("   per-row spread / peak:", " ".join(f"{float(x) / peak:.1e}" for x in dev))
    # the same utterance alone on the same engine
    o1, _ = eng.forward([ids], ref_s, forced_durations=[fd], rand_ini=ri[:1], noise=nz[:1])
    torch.cuda.synchronize()
    d1 = (wav - o1[0][None]).abs().amax(dim=1)
    print(f"  vs the utterance run alone: row 0 {float(d1[0]) / peak:.3e}, worst row {float(d1.max()) / peak:.3e} of peak")


if __name__ == "__main__":
    main()
