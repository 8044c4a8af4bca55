"""STT loader entry points (``mlx_audio/stt/utils.py:133-184``): ``load_model`` / ``load`` (strict defaults to False there)."""
from __future__ import annotations

from pathlib import Path
from typing import Any, List, Union

import numpy as np

from ..utils import base_load_model

# aliases of config.model_type / repo-name parts onto the families this package ships
MODEL_REMAPPING = {
    "whisper": "whisper",
}


def get_available_models() -> List[str]:
    d = Path(__file__).parent / "models"
    return sorted(p.name for p in d.iterdir() if p.is_dir() and not p.name.startswith("__"))


SAMPLE_RATE = 16000


def load_audio(file: Union[str, Path], sr: int = SAMPLE_RATE, from_stdin: bool = False, dtype=np.float32) -> np.ndarray:
    """Opens an audio file as a mono float waveform at ``sr`` Hz (``mlx_audio/stt/utils.py:106-130``): ``audio_io.read`` with the target rate and one
    channel.  WAV / PCM containers only in this build (``mlx_audio_amd/audio_io.py``); ``from_stdin`` is accepted for signature parity."""
    from ..audio_io import read as audio_read

    audio, _ = audio_read(file, dtype="float32", sample_rate=sr, nchannels=1)
    return audio.astype(dtype, copy=False)


def resample_audio(audio: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """Time-first resampling with the package resampler (``mlx_audio/stt/utils.py:100-103``)."""
    from ..utils import resample_audio as _resample_audio

    return _resample_audio(audio, orig_sr, target_sr, axis=0)


def load_model(model_path: Union[str, Path], lazy: bool = False, strict: bool = False, **kwargs: Any):
    return base_load_model(model_path=model_path, category="stt", model_remapping=MODEL_REMAPPING, lazy=lazy, strict=strict, **kwargs)


def load(model_path: Union[str, Path], lazy: bool = False, strict: bool = False, **kwargs: Any):
    return load_model(model_path, lazy=lazy, strict=strict, **kwargs)
