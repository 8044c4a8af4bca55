"""Codec models of the hot path, under the reference's names (``mlx_audio/codec/models/__init__.py``): ``DAC``, ``SNAC``, ``Encodec`` and ``Vocos`` are the
engines of this build (same constructor arguments as the reference classes; ``encode`` and ``decode`` since round 5 -- a checkpoint without encoder weights
decodes only); ``Mimi`` is exposed as its decoder engine (``MimiDecoder``: codes -> waveform; ``mimi.Mimi`` holds both halves).  The remaining reference exports (EcapaTdnnBackbone, MossAudioTokenizer, NemotronVoiceChatCodec,
StepAudio2Token2Wav) are outside SURVEY section 8 and raise ``ImportError`` naming that fact instead of an ``AttributeError``.  Resolved lazily so
that ``import mlx_audio_amd.codec`` stays import-light."""
import importlib

_BUILT = {
    "DAC": (".descript", "DAC"),
    "SNAC": (".snac", "SNAC"),
    "Vocos": (".vocos", "Vocos"),
    "Mimi": (".mimi", "MimiDecoder"),
    "Encodec": (".encodec", "Encodec"),
}
_NOT_BUILT = ("EcapaTdnnBackbone", "MossAudioTokenizer", "NemotronVoiceChatCodec", "StepAudio2Token2Wav")

__all__ = sorted(_BUILT)


def __getattr__(name):
    if name in _BUILT:
        mod, attr = _BUILT[name]
        return getattr(importlib.import_module(mod, __name__), attr)
    if name in _NOT_BUILT:
        raise ImportError(f"mlx_audio_amd.codec.models.{name}: this codec is outside the MI355X hot-path scope (SURVEY.md section 8) and is not built")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
