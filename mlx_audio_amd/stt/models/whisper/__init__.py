from .whisper import Model, ModelConfig, ModelDimensions  # noqa: F401

__all__ = ["Model", "ModelConfig", "ModelDimensions"]
