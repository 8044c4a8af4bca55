from .sesame import Model, Segment  # noqa: F401

__all__ = ["Model", "Segment"]
