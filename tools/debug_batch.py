"""Locates the first stage at which a ragged batch diverges from the single-utterance run (GPU debug aid)."""
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
for fuse in (False, True):
    eng.fuse_stats = fuse
    outs, durs, tb = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz, return_intermediates=True)
    torch.cuda.synchronize()
    for b in range(3):
        Fb = int(fds[b].sum())
        o1, _, t1 = eng.forward([idl[b]], refs[b:b + 1], forced_durations=[fds[b]], rand_ini=ri[b:b + 1],
                                noise=nz[b:b + 1, : 2 * Fb * 300].contiguous(), return_intermediates=True)
        torch.cuda.synchronize()
        d = float((outs[b] - o1[0]).abs().max())
        print(f"fuse={fuse} item {b} F={Fb}: audio max diff {d:.3e}")
        for k in ("d", "en", "f0", "n", "asr", "dec_in", "enc", "dec0", "dec1", "dec2", "xg", "nconv0", "nres0", "xu0", "stage0", "nconv1",
                  "nres1", "xu1", "stage1", "post"):
            if k in tb and k in t1:
                a, c = tb[k][b], t1[k][0]
                if a.dim() == 2:
                    n = min(a.shape[0], c.shape[0])
                    a, c = a[:n], c[:n]
                    # batch tensors are padded along rows: compare the single run's rows only
                    c = c[:n]
                    a = a[: c.shape[0]]
                elif a.dim() == 1:
                    a = a[: c.shape[0]]
                dd = float((a[: c.shape[0]] - c).abs().max()) if a.shape[-1] == c.shape[-1] else float("nan")
                print(f"    {k:8s} {tuple(a.shape)} vs {tuple(c.shape)} max diff {dd:.3e}  (ref max {float(c.abs().max()):.3e})")
