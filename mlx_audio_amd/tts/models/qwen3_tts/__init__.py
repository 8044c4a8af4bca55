from .config import ModelConfig  # noqa: F401
from .qwen3_tts import Model  # noqa: F401

__all__ = ["Model", "ModelConfig"]
