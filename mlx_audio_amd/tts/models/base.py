"""Result / config records of the TTS generate() protocol.

Mirrors the reference's ``tts/models/base.py:8-99`` (same field names, order and meaning) so that callers of
``model.generate()`` can switch packages without touching their code.  ``audio`` is a torch tensor on the
ROCm device instead of an ``mx.array``.
"""
from __future__ import annotations

import inspect
from dataclasses import dataclass

import torch


@dataclass
class BaseModelArgs:
    @classmethod
    def from_dict(cls, params: dict):
        """Builds the config from a checkpoint's config.json, ignoring keys the dataclass does not declare."""
        accepted = inspect.signature(cls).parameters
        return cls(**{k: v for k, v in params.items() if k in accepted})


def check_array_shape(arr) -> bool:
    """True when a 3-D conv weight is already in the MLX layout (out, K, in) rather than PyTorch's (out, in, K);
    the reference's shape heuristic (tts/models/base.py:21-34): out is the largest axis and the last two agree."""
    if len(arr.shape) != 3:
        return False
    out_channels, k_h, k_w = arr.shape
    return bool(out_channels >= k_h and out_channels >= k_w and k_h == k_w)


def format_duration(seconds: float) -> str:
    """HH:MM:SS.mmm exactly as the reference formats ``audio_duration`` (kokoro.py:337-342; minutes are not
    wrapped at 60 there, which is reproduced)."""
    return f"{int(seconds // 3600):02d}:{int(seconds // 60):02d}:{int(seconds % 60):02d}.{int((seconds % 1) * 1000):03d}"


@dataclass
class GenerationResult:
    audio: torch.Tensor
    samples: int
    sample_rate: int
    segment_idx: int
    token_count: int
    audio_duration: str
    real_time_factor: float
    prompt: dict
    audio_samples: dict
    processing_time_seconds: float
    peak_memory_usage: float
    is_streaming_chunk: bool = False
    is_final_chunk: bool = False


@dataclass
class BatchGenerationResult:
    audio: torch.Tensor  # [samples] decoded audio of one sequence
    sequence_idx: int
    samples: int
    sample_rate: int
    token_count: int
    audio_duration: str
    processing_time_seconds: float
    peak_memory_usage: float
    is_streaming_chunk: bool = False
    is_final_chunk: bool = False


def adjust_speed(audio_array, speed_factor: float) -> torch.Tensor:
    """Changes the speed of ``audio_array`` ``[samples]`` or ``[samples, channels]`` by linear-interpolation resampling (``tts/models/base.py:37-68``):
    ``int(n / speed_factor)`` output points spread evenly over ``[0, n - 1]`` (both end points kept), each a weighted mean of its two neighbours.
    ``speed_factor`` > 1 is faster.  Runs on the tensor's own device."""
    x = audio_array if isinstance(audio_array, torch.Tensor) else torch.as_tensor(audio_array)
    if not x.is_floating_point():
        x = x.to(torch.float32)
    n = x.shape[0]
    m = int(n / speed_factor)
    pos = torch.linspace(0, n - 1, m, device=x.device, dtype=torch.float32)
    lo = torch.floor(pos).to(torch.long)
    hi = torch.clamp(lo + 1, max=n - 1)
    w_hi = (pos - lo.to(torch.float32)).to(x.dtype)
    if x.dim() > 1:
        w_hi = w_hi.reshape(-1, *([1] * (x.dim() - 1)))
    return (1.0 - w_hi) * x[lo] + w_hi * x[hi]
