"""The log-mel front ends of four more dsp callers (Parakeet, Sortformer, S3 tokenizer, Voxtral Realtime: SURVEY 8(f).1) on the fused HIP kernel, under
the reference's module paths and function names, against (i) what the reference's own source files compute (tests/golden/ref_frontends.npz) and
(ii) the numpy oracle on a longer batch."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_frontends.npz"))


def _close(got, want, tol):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= tol, float(np.abs(got - want).max())


def test_parakeet_front_end():
    from mlx_audio_amd.stt.models.parakeet.audio import PreprocessArgs, log_mel_spectrogram

    a = REF["audio"]
    args = PreprocessArgs(sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann", features=80, n_fft=512, dither=0.0)
    _close(log_mel_spectrogram(a, args), REF["parakeet_per_feature"], 5e-4)
    args2 = PreprocessArgs(sample_rate=16000, normalize="global", window_size=0.025, window_stride=0.01, window="hamming", features=128, n_fft=512, dither=0.0,
                           pad_to=20000, preemph=0.0)
    _close(log_mel_spectrogram(a, args2), REF["parakeet_global_hamming_padded"], 5e-4)


def test_sortformer_features_batched():
    from mlx_audio_amd.vad.models.sortformer import extract_mel_features
    from oracle import dsp_ref

    a = REF["audio"]
    b = np.stack([a[:8000], a[4000:12000] * 0.5])
    _close(extract_mel_features(b), REF["sortformer"], 5e-4)
    _close(extract_mel_features(b[0], n_mels=128, normalize=None, pad_to=0), REF["sortformer_nonorm_128"], 5e-4)
    rng = np.random.default_rng(3)
    big = (rng.standard_normal((9, 48000)) * rng.uniform(0.01, 1.0, (9, 1))).astype(np.float32)     # 9 x 3 s in ONE launch
    _close(extract_mel_features(big), dsp_ref.sortformer_mel_features(big), 5e-4)


def test_s3_and_voxtral_front_ends():
    from mlx_audio_amd.codec.models.s3.utils import log_mel_spectrogram as s3_mel
    from mlx_audio_amd.stt.models.voxtral_realtime.audio import compute_mel_filters, compute_mel_spectrogram
    from oracle import dsp_ref

    a = REF["audio"]
    _close(s3_mel(a, padding=160), REF["s3"], 5e-5)
    fb = compute_mel_filters()
    assert fb.shape == (201, 128)
    _close(compute_mel_spectrogram(a, fb), REF["voxtral"], 5e-5)
    long = np.random.default_rng(4).standard_normal(16000 * 20).astype(np.float32) * 0.3
    _close(s3_mel(long), dsp_ref.s3_log_mel(long), 5e-5)
    _close(compute_mel_spectrogram(long), dsp_ref.voxtral_log_mel(long), 5e-5)
