"""S3 speech tokenizer front end (codec/models/s3/utils.py:8-42): Whisper's log-mel with a periodic Hann window, 128 mels and every frame kept."""
import torch

from ....frontends import whisper_style_log_mel


def log_mel_spectrogram(audio, sample_rate: int = 16_000, n_mels: int = 128, n_fft: int = 400, hop_length: int = 160, padding: int = 0) -> torch.Tensor:
    """``[L]`` samples -> ``[n_mels, n_frames]``."""
    x = torch.as_tensor(audio, dtype=torch.float32).reshape(-1)
    if padding > 0:
        x = torch.nn.functional.pad(x, (0, padding))
    return whisper_style_log_mel(x, sample_rate, n_fft, hop_length, n_mels, periodic_window=True, drop_last=False)[0].t().contiguous()
