#!/usr/bin/env python
"""Runs the reference's OWN log-mel front ends of four more dsp callers -- source files imported from /root/reference where they lie,
over the numpy stand-in for MLX (tests/golden/mlx_shim.py) -- on seeded signals and stores what they compute in tests/golden/ref_frontends.npz:

  * Parakeet        stt/models/parakeet/audio.py:39-94          log_mel_spectrogram(x, PreprocessArgs)   (NeMo: pre-emphasis, n_fft 512, ln(x + 2^-24), per-feature norm)
  * Sortformer      vad/models/sortformer/sortformer.py:36-123  extract_mel_features(waveform, ...)      (NeMo, batched, frames padded to a multiple of 16)
  * S3 tokenizer    codec/models/s3/utils.py:8-42               log_mel_spectrogram(audio, ...)          (Whisper-style, periodic Hann, 128 mels, no frame dropped)
  * Voxtral RT      stt/models/voxtral_realtime/audio.py:41-96  compute_mel_spectrogram(audio, filters)  (fixed global maximum 1.5)

Only runs in the build container: ``python tests/golden/make_frontend_fixtures.py``.  tests/test_frontends_cpu.py pins oracle/dsp_ref.py to the file.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_fixtures as M  # noqa: E402  (installs the stand-in, gives _pkg / _load)

REF = M.REF


def main():
    M.import_reference()
    dsp = sys.modules["mlx_audio.dsp"]
    utils = sys.modules["mlx_audio.utils"]
    for name in ("STR_TO_WINDOW_FN", "bartlett", "blackman", "hamming", "hanning", "mel_filters", "stft"):   # mlx_audio/utils.py re-exports these from dsp
        setattr(utils, name, getattr(dsp, name))
    for pkg, path in (("mlx_audio.stt", "stt"), ("mlx_audio.stt.models", "stt/models"), ("mlx_audio.stt.models.parakeet", "stt/models/parakeet"),
                      ("mlx_audio.stt.models.voxtral_realtime", "stt/models/voxtral_realtime"), ("mlx_audio.codec", "codec"),
                      ("mlx_audio.codec.models", "codec/models"), ("mlx_audio.codec.models.s3", "codec/models/s3")):
        if pkg not in sys.modules:
            M._pkg(pkg, f"{REF}/{path}")
    par = M._load("mlx_audio.stt.models.parakeet.audio", f"{REF}/stt/models/parakeet/audio.py")
    vox = M._load("mlx_audio.stt.models.voxtral_realtime.audio", f"{REF}/stt/models/voxtral_realtime/audio.py")
    s3 = M._load("mlx_audio.codec.models.s3.utils", f"{REF}/codec/models/s3/utils.py")
    # Sortformer's feature extraction sits at the top of the model file: execute only that part (the model classes below need the whole nn stack)
    src = open(f"{REF}/vad/models/sortformer/sortformer.py").read()
    head = src[: src.index("# FastConformer Encoder Components")]
    head = head[head.index("def preemphasis_filter"):]
    ns = {"mx": M.mx, "mel_filters": dsp.mel_filters, "hanning": dsp.hanning, "stft": dsp.stft, "_LOG_GUARD": 2 ** -24, "_NORM_CONSTANT": 1e-5}
    consts = {ln.split("=")[0].strip(): ln.split("=")[1].strip() for ln in src.splitlines() if ln.startswith(("_LOG_GUARD", "_NORM_CONSTANT"))}
    assert consts == {"_LOG_GUARD": "2**-24", "_NORM_CONSTANT": "1e-5"}, consts
    exec(compile(head, "sortformer_head", "exec"), ns)

    mx = M.mx
    rng = np.random.default_rng(11)
    out = {}
    a = (rng.standard_normal(16000 + 137) * np.linspace(0.05, 0.6, 16137)).astype(np.float32)
    out["audio"] = a
    args = par.PreprocessArgs(sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann", features=80, n_fft=512, dither=0.0)
    out["parakeet_per_feature"] = np.asarray(par.log_mel_spectrogram(mx.array(a), args))
    args2 = par.PreprocessArgs(sample_rate=16000, normalize="global", window_size=0.025, window_stride=0.01, window="hamming", features=128, n_fft=512, dither=0.0,
                               pad_to=20000, preemph=0.0)
    out["parakeet_global_hamming_padded"] = np.asarray(par.log_mel_spectrogram(mx.array(a), args2))
    b = np.stack([a[:8000], a[4000:12000] * 0.5])
    out["sortformer"] = np.asarray(ns["extract_mel_features"](mx.array(b)))
    out["sortformer_nonorm_128"] = np.asarray(ns["extract_mel_features"](mx.array(b[0]), n_mels=128, normalize=None, pad_to=0))
    out["s3"] = np.asarray(s3.log_mel_spectrogram(mx.array(a), padding=160))
    fb = vox.compute_mel_filters()
    out["voxtral"] = np.asarray(vox.compute_mel_spectrogram(mx.array(a), mx.array(fb)))
    for k, v in out.items():
        print(k, v.shape, v.dtype, float(np.abs(v).max()))
    np.savez_compressed(os.path.join(HERE, "ref_frontends.npz"), **out)


if __name__ == "__main__":
    main()
