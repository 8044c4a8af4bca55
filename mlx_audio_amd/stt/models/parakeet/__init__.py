from .audio import PreprocessArgs, log_mel_spectrogram  # noqa: F401
