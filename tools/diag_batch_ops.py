#!/usr/bin/env python
"""Diagnostic (finer than diag_batch_rows.py): B IDENTICAL utterances through one engine call with every ``ops.*`` launch wrapped -- after each
launch the outputs' rows are compared with row 0 (they must be equal: every row walks the same tiles of the same kernel).  Prints the launches
in call order whose outputs differ between rows, and for each the same check on its INPUTS (so the first line with clean inputs and a dirty
output names the kernel).  ``--poison`` fills the caching allocator's free blocks with NaN first (a read of uninitialised memory then shows).

    python tools/diag_batch_ops.py --precision 2 --batch 64 [--poison] [--front-only]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def spread(t, B):
    if not isinstance(t, torch.Tensor) or t.dim() == 0 or t.shape[0] != B or not t.is_floating_point():
        return None
    v = t.reshape(B, -1)
    ok = torch.isfinite(v).all()
    d = (v - v[0:1]).abs().amax()
    s = v[0].abs().amax()
    return float(d), float(s), bool(ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", type=int, default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--max-lines", type=int, default=60)
    args = ap.parse_args()
    from mlx_audio_amd import ops
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
    # warm-up (allocator state like a steady-state step), then the instrumented call
    eng.forward([ids] * B, ref_s.repeat(B, 1), forced_durations=[fd] * B, rand_ini=ri, noise=nz)
    torch.cuda.synchronize()
    if args.poison:
        free, _ = torch.cuda.mem_get_info()
        big = torch.full((int(min(free * 0.5, 60e9)) // 4,), float("nan"), device="cuda")
        del big
        torch.cuda.synchronize()

    log = []
    counter = [0]

    def wrap(name, outs_of):
        fn = getattr(ops, name)

        def w(*a, **k):
            ins = [(f"arg{i}", v) for i, v in enumerate(a)] + list(k.items())
            in_sp = [(n, spread(v, B)) for n, v in ins]
            if name == "conv_gemm":   # tuples (pre = (scale, shift))
                for n, v in list(k.items()):
                    if isinstance(v, tuple):
                        in_sp += [(f"{n}[{j}]", spread(x, B)) for j, x in enumerate(v)]
            r = fn(*a, **k)
            torch.cuda.synchronize()
            outs = outs_of(a, k, r)
            rec = dict(i=counter[0], name=name, outs=[(n, tuple(v.shape), spread(v, B)) for n, v in outs if isinstance(v, torch.Tensor)],
                       ins=[(n, s) for n, s in in_sp if s is not None])
            if name == "conv_gemm":
                pc = a[1]
                rec["what"] = f"cin={getattr(pc, 'cin', '?')} cout={getattr(pc, 'cout', '?')} k={getattr(pc, 'k', '?')} kw={sorted(kk for kk in k if k[kk] is not None and kk not in ('pre',))}"
            log.append(rec)
            counter[0] += 1
            return r
        setattr(ops, name, w)

    wrap("conv_gemm", lambda a, k, r: [("y", a[2])] + ([("stats", k["stats"])] if k.get("stats") is not None else []))
    wrap("lstm_bidir", lambda a, k, r: [("out", a[3])])
    wrap("adain_coef", lambda a, k, r: [("scale", r[0]), ("shift", r[1])])
    wrap("adain_from_partials", lambda a, k, r: [("scale", r[0]), ("shift", r[1])])
    wrap("adain_pool_up2", lambda a, k, r: [("pooled", a[6])])
    wrap("layernorm", lambda a, k, r: [("y", a[1])])
    wrap("gather_rows", lambda a, k, r: [("y", a[2])])
    wrap("sine_source", lambda a, k, r: [("har_src", r)])
    wrap("conv1d_c1_k3s2", lambda a, k, r: [("dst", a[3])])

    outs, _ = eng.forward([ids] * B, ref_s.repeat(B, 1), forced_durations=[fd] * B, rand_ini=ri, noise=nz)
    torch.cuda.synchronize()
    n_bad = 0
    print(f"precision {eng.precision}, {B} identical utterances, {len(log)} wrapped launches; launches whose outputs differ between rows (first {args.max_lines}):")
    for rec in log:
        bad_out = [(n, sh, s) for n, sh, s in rec["outs"] if s is not None and (s[0] > 0 or not s[2])]
        if not bad_out:
            continue
        n_bad += 1
        if n_bad > args.max_lines:
            continue
        bad_in = [(n, s) for n, s in rec["ins"] if s[0] > 0 or not s[2]]
        o = "; ".join(f"{n}{sh}: {s[0]:.2e}/{s[1]:.2e}{'' if s[2] else ' NONFINITE'}" for n, sh, s in bad_out)
        i = "CLEAN INPUTS" if not bad_in else "dirty inputs: " + ", ".join(f"{n} {s[0]:.1e}{'' if s[2] else ' NONFINITE'}" for n, s in bad_in)
        print(f"  #{rec['i']:3d} {rec['name']:20s} {rec.get('what', '')}\n        out {o}\n        {i}")
    print(f"{n_bad} of {len(log)} launches have row-dependent outputs")
    wav = torch.stack(outs)
    print(f"waveform spread {float((wav - wav[0:1]).abs().max()):.3e} (peak {float(wav[0].abs().max()):.3f})")


if __name__ == "__main__":
    main()
