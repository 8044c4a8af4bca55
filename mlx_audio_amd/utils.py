"""Model loading shared by the kinds (mirrors the loader contract of ``mlx_audio/utils.py:270-420,832``).

Resolution: ``config["model_type"]`` -> ``config["architecture"]`` -> first part of the repo name; aliases through the
kind's ``MODEL_REMAPPING``, then repo-name parts; the architecture module is ``mlx_audio_amd.<kind>.models.<type>`` and
must expose ``Model`` and optionally ``ModelConfig.from_dict``.  Errors keep the reference's types: missing local path
/ weights -> ``FileNotFoundError``, unknown type -> ``ValueError``, missing optional dependency -> ``ImportError`` with a
pip hint.  Weights are ``*.safetensors`` read as torch tensors; there is no network in this build's test environment,
so hub downloads are attempted only when ``huggingface_hub`` can reach the hub.
"""
from __future__ import annotations

import glob
import importlib
import json
import logging
import re
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union


def get_model_name_parts(model_path: Union[str, Path]) -> List[str]:
    return [s for s in re.split(r"[-_. ]+", Path(str(model_path)).name.lower()) if s]


def get_model_path(path_or_hf_repo: str, revision: Optional[str] = None, force_download: bool = False,
                   allow_patterns: Optional[List[str]] = None) -> Path:
    p = Path(path_or_hf_repo)
    if p.exists():
        return p
    if str(path_or_hf_repo).startswith((".", "/", "~")):
        raise FileNotFoundError(f"Model path {path_or_hf_repo} does not exist")
    try:
        from huggingface_hub import snapshot_download

        return Path(snapshot_download(path_or_hf_repo, revision=revision, force_download=force_download,
                                      allow_patterns=allow_patterns or ["*.json", "*.safetensors", "*.txt", "voices/*"]))
    except Exception as e:
        raise FileNotFoundError(f"Model {path_or_hf_repo!r} is not a local path and could not be fetched: {e}") from e


def load_config(model_path: Union[str, Path]) -> dict:
    cfg = Path(model_path) / "config.json"
    if not cfg.exists():
        raise FileNotFoundError(f"Config not found at {model_path}")
    with open(cfg, encoding="utf-8") as f:
        return json.load(f)


def load_weights(model_path: Union[str, Path]) -> Dict[str, "object"]:
    files = sorted(glob.glob(str(Path(model_path) / "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No safetensors found in {model_path}")
    from safetensors.torch import load_file

    weights = {}
    for f in files:
        weights.update(load_file(f))
    return weights


def get_model_class(model_type: str, model_name: Optional[List[str]], category: str, model_remapping: dict):
    mapped = model_remapping.get(model_type)
    models_dir = Path(__file__).parent / category / "models"
    available = [p.name for p in models_dir.iterdir() if p.is_dir() and not p.name.startswith("__")] if models_dir.is_dir() else []
    if model_name is not None and mapped != model_type:
        for part in model_name:
            if part in available:
                model_type = part
            if part in model_remapping:
                model_type = model_remapping[part]
                break
    elif mapped is not None:
        model_type = mapped
    module_path = f"mlx_audio_amd.{category}.models.{model_type}"
    try:
        arch = importlib.import_module(module_path)
    except ImportError as e:
        if e.name != module_path:
            raise ImportError(f"\nMissing dependency while loading {model_type}: {e}\nPlease install it using: pip install {e.name}") from e
        msg = f"Model type {model_type} not supported for {category}."
        logging.error(msg)
        raise ValueError(msg)
    if not hasattr(arch, "Model"):  # a family directory that ships engines but no Model: the reference's error for an unknown type
        msg = f"Model type {model_type} not supported for {category}."
        logging.error(msg)
        raise ValueError(msg)
    return arch, model_type


def base_load_model(model_path: Union[str, Path], category: str, model_remapping: dict, lazy: bool = False, strict: bool = False, **kwargs):
    model_name = kwargs.pop("model_name_parts", None)
    model_type = kwargs.pop("model_type", None)
    allow_patterns = kwargs.pop("allow_patterns", None)
    if isinstance(model_path, str):
        model_name = model_name or get_model_name_parts(model_path)
        model_path = get_model_path(model_path, revision=kwargs.get("revision"), force_download=kwargs.get("force_download", False),
                                    allow_patterns=allow_patterns)
    elif isinstance(model_path, Path):
        model_name = model_name or get_model_name_parts(model_path)
        if not model_path.exists():
            raise FileNotFoundError(f"Model path {model_path} does not exist")
    else:
        raise ValueError(f"Invalid model path type: {type(model_path)}")
    config = load_config(model_path)
    config["model_path"] = str(model_path)
    model_type = model_type or config.get("model_type") or config.get("architecture") or (model_name[0].lower() if model_name else None)
    arch, model_type = get_model_class(model_type=model_type, model_name=model_name, category=category, model_remapping=model_remapping)
    model_config = arch.ModelConfig.from_dict(config) if hasattr(arch, "ModelConfig") else config
    extra = {k: kwargs[k] for k in ("device", "precision") if k in kwargs}
    model = arch.Model(model_config, **extra)
    weights = load_weights(model_path)
    if hasattr(model, "sanitize"):
        weights = model.sanitize(weights)
    model.model_path = str(model_path)
    model.load_weights(list(weights.items()), strict=strict)
    model.eval()
    if hasattr(arch.Model, "post_load_hook"):
        model = arch.Model.post_load_hook(model, model_path)
    return model


def load_model(model_name: str, **kwargs):
    """Kind-agnostic entry point (``mlx_audio.utils.load_model``, utils.py:832): classifies the model from its config /
    name and dispatches to the kind's loader."""
    from . import registry

    path = get_model_path(model_name) if isinstance(model_name, str) else model_name
    cfg = load_config(path)
    kind = registry.classify_model(cfg.get("model_type") or cfg.get("architecture") or "", str(model_name))
    if kind is None:
        raise ValueError(f"Model type {cfg.get('model_type')} not supported.")
    return importlib.import_module(f"mlx_audio_amd.{kind}.utils").load_model(Path(path), **kwargs)


def resample_audio(audio, orig_sample_rate: int, sample_rate: int, axis: int = -1):
    """``mlx_audio.utils.resample_audio`` (utils.py:541-578): polyphase resampling with the reference's ``kaiser_best`` filter; the return type matches
    the input type (numpy array or torch tensor)."""
    from .resample import resample_audio as _impl

    return _impl(audio, orig_sample_rate, sample_rate, axis=axis)


def audio_volume_normalize(audio, coeff: float = 0.2):
    """``mlx_audio.utils.audio_volume_normalize`` (utils.py:477-516): quiet clips (peak < 0.1) are first scaled to a peak of 0.1; then the mean of the
    90th..99th percentile magnitudes above 0.01 is brought to ``coeff`` (gain clipped to [0.1, 10]) and the result to a peak of at most 1.  numpy in,
    numpy out."""
    import numpy as np

    temp = np.sort(np.abs(audio))
    if temp[-1] < 0.1:
        audio = audio / max(temp[-1], 1e-3) * 0.1
    temp = temp[temp > 0.01]
    L = temp.shape[0]
    if L <= 10:
        return audio
    volume = np.mean(temp[int(0.9 * L): int(0.99 * L)])
    audio = audio * np.clip(coeff / volume, a_min=0.1, a_max=10)
    peak = np.max(np.abs(audio))
    return audio / peak if peak > 1 else audio


def random_select_audio_segment(audio, length: int):
    """``mlx_audio.utils.random_select_audio_segment`` (utils.py:519-538): a random window of ``length`` samples (zero-padded when the clip is shorter)."""
    import random

    import numpy as np

    if audio.shape[0] < length:
        audio = np.pad(audio, (0, int(length - audio.shape[0])))
    start = random.randint(0, audio.shape[0] - length)
    return audio[start: int(start + length)]


def trim_silence(audio, top_db: float = 20, frame_length: int = 2048, hop_length: int = 512):
    """``mlx_audio.utils.trim_silence`` (utils.py:580-617): frames whose RMS is within ``top_db`` of the loudest frame bound the kept span."""
    import numpy as np
    import torch

    a = audio.detach().cpu().numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
    n_frames = 1 + (len(a) - frame_length) // hop_length
    if n_frames <= 0:
        return audio
    rms = np.array([np.sqrt(np.mean(a[i * hop_length: i * hop_length + frame_length] ** 2)) for i in range(n_frames)])
    rms_db = 20 * np.log10(np.maximum(rms, 1e-10))
    keep = np.where(rms_db >= np.max(rms_db) - top_db)[0]
    if len(keep) == 0:
        return audio
    out = a[int(keep[0]) * hop_length: min((int(keep[-1]) + 1) * hop_length + frame_length, len(a))].astype(np.float32, copy=False)
    return torch.from_numpy(np.ascontiguousarray(out)).to(audio.device) if isinstance(audio, torch.Tensor) else out


def load_audio(audio, sample_rate: int = 24000, length: Optional[int] = None, volume_normalize: bool = False, segment_duration: Optional[int] = None):
    """``mlx_audio.utils.load_audio`` (utils.py:620-676): a path is read (mono, resampled to ``sample_rate``, float32), optionally cut to a random
    segment, volume-normalised, padded / truncated to ``length``; a tensor (the reference: an ``mx.array``) is returned as is.  Returns a float32
    ``torch.Tensor`` on the host; the engines move it to the device."""
    import os

    import numpy as np
    import torch

    if isinstance(audio, torch.Tensor):
        return audio
    if not isinstance(audio, str):
        raise TypeError(f"audio must be str or torch.Tensor, got {type(audio)}")
    from .audio_io import read as audio_read

    if not os.path.exists(audio):
        raise FileNotFoundError(f"Audio file not found: {audio}")
    samples, _ = audio_read(audio, dtype="float32", sample_rate=sample_rate, nchannels=1)
    if segment_duration is not None:
        samples = random_select_audio_segment(samples, int(sample_rate * segment_duration))
    if volume_normalize:
        samples = audio_volume_normalize(samples)
    if length is not None:
        samples = samples[:length] if samples.shape[0] > length else np.pad(samples, (0, int(length - samples.shape[0])))
    return torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32))


# ``mlx_audio.utils`` re-exports the dsp entry points (utils.py:31-40; the reference's tests import them from here: tests/test_dsp.py:30-38).  Resolved
# lazily so that ``import mlx_audio_amd.utils`` (loader, registry users) does not pull the dsp module in.
_DSP_REEXPORTS = ("STR_TO_WINDOW_FN", "bartlett", "blackman", "hamming", "hanning", "istft", "mel_filters", "stft")


def __getattr__(name):
    if name in _DSP_REEXPORTS:
        from . import dsp

        return getattr(dsp, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
