"""The numpy restatements of four more log-mel front ends (oracle/dsp_ref.py: Parakeet, Sortformer, S3 tokenizer, Voxtral Realtime) against what
the reference's own source files compute over the MLX stand-in (tests/golden/ref_frontends.npz, written by tests/golden/make_frontend_fixtures.py)."""
import os

import numpy as np

from oracle import dsp_ref

REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_frontends.npz"))


def _close(got, want, tol):
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max() <= tol, float(np.abs(got - want).max())


def test_parakeet_restatement_matches_reference_run():
    a = REF["audio"]
    _close(dsp_ref.parakeet_log_mel(a), REF["parakeet_per_feature"], 2e-4)
    _close(dsp_ref.parakeet_log_mel(a, normalize="global", window="hamming", features=128, pad_to=20000, preemph=0.0), REF["parakeet_global_hamming_padded"], 2e-4)


def test_sortformer_restatement_matches_reference_run():
    a = REF["audio"]
    b = np.stack([a[:8000], a[4000:12000] * 0.5])
    got = dsp_ref.sortformer_mel_features(b)
    assert got.shape == (2, 80, 64) and np.all(got[:, :, 51:] == 0.0)     # 51 frames padded to a multiple of 16
    _close(got, REF["sortformer"], 2e-4)
    _close(dsp_ref.sortformer_mel_features(b[0], n_mels=128, normalize=None, pad_to=0), REF["sortformer_nonorm_128"], 2e-4)


def test_s3_and_voxtral_restatements_match_reference_run():
    a = REF["audio"]
    _close(dsp_ref.s3_log_mel(a, padding=160), REF["s3"], 2e-5)
    _close(dsp_ref.voxtral_log_mel(a), REF["voxtral"], 2e-5)


def test_voxtral_mel_rejects_a_foreign_filter_bank():
    """``compute_mel_spectrogram`` applies its own device copy of the 16 kHz / 0-8000 Hz Slaney bank; a caller's different bank must fail loudly, not be
    silently replaced (the check runs before any device work)."""
    import pytest
    import torch

    from mlx_audio_amd.stt.models.voxtral_realtime import audio as A

    fb = A.compute_mel_filters(128, 400, 16000)
    with pytest.raises(ValueError, match="differs from the built-in"):
        A.compute_mel_spectrogram(torch.zeros(1600), fb * 1.01)
    with pytest.raises(ValueError, match="differs from the built-in"):
        A.compute_mel_spectrogram(torch.zeros(1600), torch.from_numpy(fb[:, ::-1].copy()))
