import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from mlx_audio_amd.tts.models.kokoro import synthetic as S
from mlx_audio_amd.tts.models.kitten_tts import synthetic as KS
from mlx_audio_amd.tts.models.kitten_tts.engine import KittenEngine
from mlx_audio_amd.tts.models.kokoro.engine import KokoroEngine
def rms(a,b): a=np.asarray(a,dtype=np.float64); b=np.asarray(b,dtype=np.float64); return float(np.sqrt(((a-b)**2).mean())/np.sqrt((b**2).mean()))
for name in ('kitten_tiny_plain','kokoro_tiny'):
    fx=np.load(f'/root/repo/tests/golden/ref_{name}.npz')
    if 'kitten' in name:
        cfg=KS.tiny_config(); w=KS.make_kitten_weights(cfg, seed=int(fx['seed_w'])); E=KittenEngine
    else:
        cfg=S.tiny_config(); w=S.make_kokoro_weights(cfg, seed=int(fx['seed_w'])); E=KokoroEngine
    ids=S.make_phoneme_ids(int(fx['n_phon']), seed=int(fx['seed_ids'])); ref_s=S.make_voice_pack()[len(ids)-3]
    L=fx['audio'].shape[1]
    rng=np.random.default_rng(int(fx['seed_rng'])); ri=rng.uniform(size=(1,9)).astype(np.float32); nz=rng.standard_normal((1,L,9)).astype(np.float32)
    for pdt, prec in ((torch.float32, 2), (torch.float32, 4), (torch.bfloat16, 2)):
        eng=E(w,cfg,param_dtype=pdt,precision=prec)
        outs,durs,tg=eng.forward([ids],ref_s,speed=float(fx['speed']),rand_ini=torch.from_numpy(ri),noise=torch.from_numpy(nz),return_intermediates=True)
        torch.cuda.synchronize()
        print(name, pdt, 'precision', prec, 'dur eq', np.array_equal(durs[0].cpu().numpy(), fx['pred_dur']), 'd', rms(tg['d'][0].cpu().numpy(), fx['d'][0]), 'f0', rms(tg['f0'].cpu().numpy(), fx['f0']), 'n', rms(tg['n'].cpu().numpy(), fx['n']), 'asr', rms(tg['asr'][0].cpu().numpy().T, fx['asr'][0]))
        outs2,_,tg2=eng.forward([ids],ref_s,forced_durations=[torch.from_numpy(fx['pred_dur'])],rand_ini=torch.from_numpy(ri),noise=torch.from_numpy(nz),overrides=dict(f0=torch.from_numpy(fx['f0']),n=torch.from_numpy(fx['n'])),return_intermediates=True)
        torch.cuda.synchronize()
        a=outs2[0].cpu().numpy(); b=fx['audio'][0]
        print('   audio relrms', rms(a,b), 'maxabs', float(np.abs(a-b).max()), 'peak', float(np.abs(b).max()), 'snr', 10*np.log10((b.astype(np.float64)**2).sum()/((a-b).astype(np.float64)**2).sum()))
