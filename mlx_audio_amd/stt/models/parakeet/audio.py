"""Parakeet's audio front end (stt/models/parakeet/audio.py): the same ``PreprocessArgs`` and ``log_mel_spectrogram(x, args)`` on the MI355X --
pre-emphasis + one fused STFT -> |X|^2 -> mel -> ln(. + guard) kernel + the normalisation.  The Parakeet model itself is outside SURVEY 8(a)."""
from dataclasses import dataclass

import torch

from ....frontends import nemo_log_mel, per_feature_norm


@dataclass
class PreprocessArgs:
    sample_rate: int
    normalize: str
    window_size: float
    window_stride: float
    window: str
    features: int
    n_fft: int
    dither: float
    pad_to: int = 0
    pad_value: float = 0
    preemph: float = 0.97
    log_zero_guard_value: float = 2 ** -24

    @property
    def win_length(self) -> int:
        return int(self.window_size * self.sample_rate)

    @property
    def hop_length(self) -> int:
        return int(self.window_stride * self.sample_rate)


def log_mel_spectrogram(x, args: PreprocessArgs) -> torch.Tensor:
    """``[L]`` samples -> ``[1, n_frames, features]`` (parakeet/audio.py:39-94)."""
    x = torch.as_tensor(x, dtype=torch.float32).reshape(-1)
    if args.pad_to > 0 and x.shape[-1] < args.pad_to:
        x = torch.nn.functional.pad(x, (0, args.pad_to - x.shape[-1]), value=float(args.pad_value))
    y = nemo_log_mel(x, args.sample_rate, args.n_fft, args.hop_length, args.win_length, args.features, args.window,
                     float(getattr(args, "preemph", 0.97)), float(args.log_zero_guard_value))[0]       # [frames, features]
    if args.normalize == "per_feature":
        y = per_feature_norm(y, dim=0)
    else:
        y = (y - y.mean()) / (y.std(unbiased=False) + 1e-5)
    return y[None]
