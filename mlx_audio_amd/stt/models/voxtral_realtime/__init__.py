from .audio import compute_mel_filters, compute_mel_spectrogram  # noqa: F401
