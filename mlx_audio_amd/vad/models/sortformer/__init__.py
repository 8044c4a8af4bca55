from .sortformer import extract_mel_features, preemphasis_filter  # noqa: F401
