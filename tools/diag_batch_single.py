#!/usr/bin/env python3
"""Where does a batch-vs-single difference sit?  (round 4, call 18: with split groups of >= 2 steps test_kokoro_batch_equals_single fails by 2.2 of a
17 peak.)  Prints, per item: the max |diff|, the share of samples above 1e-3 of the peak, the first / last such sample, and the same for F0 / N
handed back through return_intermediates when the engine offers them."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_audio_amd.tts.models.kokoro import synthetic as S  # noqa: E402
from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine  # noqa: E402

eng = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG)
voice = S.make_voice_pack()
idl = [S.make_phoneme_ids(n, seed=10 + n) for n in (12, 25, 7)]
refs = torch.cat([voice[len(i) - 3] for i in idl], 0)
fds = [S.forced_durations(len(i), 3 * len(i), seed=len(i)) for i in idl]
Fm = max(int(f.sum()) for f in fds)
rng = np.random.default_rng(5)
ri = torch.from_numpy(rng.uniform(size=(3, 9)).astype(np.float32))
nz = torch.from_numpy(rng.standard_normal((3, 2 * Fm * 300, 9)).astype(np.float32))
kw = {}
try:
    outs, durs, trb = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz, return_intermediates=True)
    kw = dict(return_intermediates=True)
except TypeError:
    outs, durs = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz)
    trb = None
torch.cuda.synchronize()
for b in range(3):
    Fb = int(fds[b].sum())
    r = eng.forward([idl[b]], refs[b:b + 1], forced_durations=[fds[b]], rand_ini=ri[b:b + 1], noise=nz[b:b + 1, : 2 * Fb * 300].contiguous(), **kw)
    torch.cuda.synchronize()
    o1 = r[0][0]
    d = (outs[b] - o1).abs().cpu()
    peak = float(o1.abs().max())
    big = (d > 1e-3 * peak).nonzero().flatten()
    print(f"item {b}: max {float(d.max()):.3e} peak {peak:.2f} share>1e-3 {big.numel() / d.numel():.4f} first {int(big[0]) if big.numel() else -1} last {int(big[-1]) if big.numel() else -1} n {d.numel()}")
    if trb is not None:
        tr1 = r[2]
        for k in sorted(set(trb) & set(tr1)):
            x, y = trb[k], tr1[k]
            if not (torch.is_tensor(x) and torch.is_tensor(y)) or x.dim() != y.dim() or x.shape[0] != 3:
                continue
            x = x[b:b + 1]
            sl = tuple(slice(0, min(p, q)) for p, q in zip(x.shape, y.shape))
            dd = (x[sl].float().cpu() - y[sl].float().cpu()).abs()
            print(f"   {k}: max diff {float(dd.max()):.3e} of {float(y[sl].float().abs().max()):.3e}  at {tuple(int(v) for v in (dd == dd.max()).nonzero()[0])} shape {tuple(dd.shape)}")
