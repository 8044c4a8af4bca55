"""TTS loader entry points (``mlx_audio/tts/utils.py:100-155``): ``load_model`` / ``load``."""
from __future__ import annotations

from pathlib import Path
from typing import Any, List, Union

from ..utils import base_load_model

# aliases of config.model_type / repo-name parts onto the families this package ships
MODEL_REMAPPING = {
    "kokoro": "kokoro",
    "kokoro_82m": "kokoro",
    "styletts2": "kokoro",
    "kitten": "kitten_tts",
    "kitten_tts": "kitten_tts",
    "qwen3_tts": "qwen3_tts",
    "csm": "sesame",
    "marvis": "sesame",
    "sesame": "sesame",
}


def get_available_models() -> List[str]:
    d = Path(__file__).parent / "models"
    return sorted(p.name for p in d.iterdir() if p.is_dir() and not p.name.startswith("__"))


def load_model(model_path: Union[str, Path], lazy: bool = False, strict: bool = True, **kwargs: Any):
    return base_load_model(model_path=model_path, category="tts", model_remapping=MODEL_REMAPPING, lazy=lazy, strict=strict, **kwargs)


def load(model_path: Union[str, Path], lazy: bool = False, strict: bool = True, **kwargs: Any):
    return load_model(model_path, lazy=lazy, strict=strict, **kwargs)
