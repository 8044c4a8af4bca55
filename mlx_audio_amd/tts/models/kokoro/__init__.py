from .kokoro import Model, ModelConfig  # noqa: F401
from .pipeline import KokoroPipeline  # noqa: F401

__all__ = ["KokoroPipeline", "Model", "ModelConfig"]
