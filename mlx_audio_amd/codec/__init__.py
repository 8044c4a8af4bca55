"""``mlx_audio.codec`` surface (``mlx_audio/codec/__init__.py``): the codec classes, resolved lazily from ``.models``."""
from . import models as _models

__all__ = list(_models.__all__)


def __getattr__(name):
    return getattr(_models, name)
