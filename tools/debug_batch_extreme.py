"""Locates the first stage at which the SHORT item of an extreme ragged batch (512 tokens next to 4) diverges from its single-utterance run."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_audio_amd.tts.models.kokoro import synthetic as S  # noqa: E402
from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine  # noqa: E402

eng = KokoroEngine(S.make_kokoro_weights(), S.KOKORO_CONFIG)
voice = S.make_voice_pack()
for nlong in (510, 60, 12):
    idl = [S.make_phoneme_ids(nlong, seed=3), S.make_phoneme_ids(2, seed=4)]
    refs = torch.cat([voice[len(i) - 3] for i in idl], 0)
    fds = [torch.ones(len(i), dtype=torch.int32) for i in idl]
    Fs = [int(f.sum()) for f in fds]
    rng = np.random.default_rng(11)
    ri = torch.from_numpy(rng.uniform(size=(2, 9)).astype(np.float32))
    nz = torch.from_numpy(rng.standard_normal((2, 2 * max(Fs) * 300, 9)).astype(np.float32))
    outs, durs, tb = eng.forward(idl, refs, forced_durations=fds, rand_ini=ri, noise=nz, return_intermediates=True)
    torch.cuda.synchronize()
    b = 1
    o1, _, t1 = eng.forward([idl[b]], refs[b:b + 1], forced_durations=[fds[b]], rand_ini=ri[b:b + 1], noise=nz[b:b + 1, : 2 * Fs[b] * 300].contiguous(),
                            return_intermediates=True)
    torch.cuda.synchronize()
    print(f"long item {nlong + 2} tokens; short item F={Fs[b]}: audio max diff {float((outs[b] - o1[0]).abs().max()):.3e} (peak {float(o1[0].abs().max()):.3f})")
    for k in ("bert", "d", "en", "f0", "n", "asr", "dec_in", "enc", "dec0", "dec1", "dec2", "xg", "har_src", "har", "nconv0", "nres0", "xu0", "stage0", "nconv1",
              "nres1", "xu1", "stage1", "post"):
        if k in tb and k in t1:
            a, c = tb[k][b], t1[k][0]
            if a.dim() == 2:
                n = min(a.shape[0], c.shape[0])
                a, c = a[:n], c[:n]
            elif a.dim() == 1:
                a = a[: c.shape[0]]
            dd = float((a - c).abs().max()) if a.shape == c.shape else float("nan")
            print(f"    {k:8s} {tuple(a.shape)} vs {tuple(c.shape)} max diff {dd:.3e}  (ref max {float(c.abs().max()):.3e})")
