"""Stage-by-stage comparison of the HIP Kokoro path with the CPU oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mlx_audio_amd.tts.models.kokoro import synthetic as S
from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine
from oracle.kokoro_ref import KokoroRef

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
w = S.make_kokoro_weights()
eng = KokoroEngine(w, S.KOKORO_CONFIG)
ref = KokoroRef(S.make_kokoro_weights(), S.KOKORO_CONFIG)
ids = S.make_phoneme_ids(T - 2, seed=5)
ref_s = S.make_voice_pack()[len(ids) - 3]
fd = S.forced_durations(T, 3 * T)
F = int(fd.sum())
rng = np.random.default_rng(7)
ri = rng.uniform(size=(1, 9)).astype(np.float32); nz = rng.standard_normal((1, 2 * F * 300, 9)).astype(np.float32)
a_ref, _, tr = ref.forward(ids, ref_s, pred_dur=fd, rand_ini=ri, noise=nz, return_intermediates=True)
tf = "--free" not in sys.argv
outs, durs, tg = eng.forward([ids], ref_s, forced_durations=[fd], rand_ini=torch.from_numpy(ri), noise=torch.from_numpy(nz), return_intermediates=True,
                             overrides=(dict(f0=tr["f0"], n=tr["n"], **({"har": tr["har"].transpose(1, 2)} if "--har" in sys.argv else {})) if tf else None))
hd = (tg["har"].cpu() - tr["har"].transpose(1, 2)).abs()
print("har flips (frame, channel):", [(int(i[1]), int(i[2])) for i in (hd > 1.0).nonzero()][:20], "n_frames", hd.shape[1])
torch.cuda.synchronize()
def cmp(name, g, r, ncl=True):
    g = g.detach().cpu().double()
    r = r.detach().double()
    if ncl and r.dim() == 3:
        r = r.transpose(1, 2)
    if g.shape != r.shape:
        print(f"{name:10s} SHAPE MISMATCH got {tuple(g.shape)} ref {tuple(r.shape)}"); return
    e = (g - r).abs()
    print(f"{name:10s} shape={tuple(g.shape)} ref_absmax={r.abs().max():.4g} max_err={e.max():.3e} rel={e.max()/(r.abs().max()+1e-30):.3e} mean_err={e.mean():.3e}")
cmp("d", tg["d"], tr["d"], ncl=False)
cmp("f0", tg["f0"], tr["f0"], ncl=False)
cmp("n", tg["n"], tr["n"], ncl=False)
for k in ["dec_in", "enc", "dec0", "dec1", "dec2", "dec3", "har_src", "har", "nconv0", "nres0", "xu0", "stage0", "nconv1", "nres1", "xu1", "stage1", "post"]:
    if k in tg and k in tr:
        cmp(k, tg[k], tr[k], ncl=(k != "har_src"))
    else:
        print(k, "missing", k in tg, k in tr)
cmp("audio", outs[0][None], a_ref, ncl=False)
