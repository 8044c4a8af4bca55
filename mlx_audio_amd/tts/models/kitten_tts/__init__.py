from .kitten_tts import Model, ModelConfig  # noqa: F401

__all__ = ["Model", "ModelConfig"]
