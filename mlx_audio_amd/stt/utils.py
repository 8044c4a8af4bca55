"""STT loader entry points (``mlx_audio/stt/utils.py:133-184``): ``load_model`` / ``load`` (strict defaults to False there)."""
from __future__ import annotations

from pathlib import Path
from typing import Any, List, Union

from ..utils import base_load_model

# aliases of config.model_type / repo-name parts onto the families this package ships
MODEL_REMAPPING = {
    "whisper": "whisper",
}


def get_available_models() -> List[str]:
    d = Path(__file__).parent / "models"
    return sorted(p.name for p in d.iterdir() if p.is_dir() and not p.name.startswith("__"))


def load_model(model_path: Union[str, Path], lazy: bool = False, strict: bool = False, **kwargs: Any):
    return base_load_model(model_path=model_path, category="stt", model_remapping=MODEL_REMAPPING, lazy=lazy, strict=strict, **kwargs)


def load(model_path: Union[str, Path], lazy: bool = False, strict: bool = False, **kwargs: Any):
    return load_model(model_path, lazy=lazy, strict=strict, **kwargs)
