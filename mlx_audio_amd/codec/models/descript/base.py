"""The file API of the Descript codec: ``DACFile`` and ``CodecMixin.compress / decompress`` (``mlx_audio/codec/models/descript/base.py:13-231``), on the
MI355X engine's ``encode`` / ``decode``.

What the reference's mixin EFFECTIVELY does is restated here, not its mechanism: its ``padding`` switch, ``get_delay`` and ``get_output_length`` walk
``self.modules()`` for ``nn.Conv1d`` / ``nn.ConvTranspose1d`` instances, of which a DAC built from ``WNConv1d`` / ``WNConvTranspose1d`` has none -- so the
delay is 0, the output length of n samples is n, and switching the padding off changes nothing (pinned by running the reference:
``tests/golden/ref_dac_compress.npz``).  ``compress`` is therefore: loudness-normalise to ``normalize_db``, cut the signal into windows of
``win_duration`` seconds rounded up to whole hops (one window when the signal is shorter), zero-pad the last one, encode each, concatenate the codes
along time; ``decompress``: decode ``chunk_length`` frames at a time, concatenate, undo the normalisation.  Nothing is trimmed (neither does the reference)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch

SUPPORTED_VERSIONS = ["1.0.0"]


@dataclass
class DACFile:
    codes: torch.Tensor          # int [1, n_codebooks, frames]
    chunk_length: int            # frames per encoded window
    original_length: float       # seconds (the reference stores the DURATION under this name)
    input_db: float
    channels: int
    sample_rate: int
    padding: bool
    dac_version: str

    def save(self, path) -> Path:
        """``.dac`` = ``np.save`` of {codes uint16, metadata} (base.py:26-43)."""
        meta = dict(input_db=float(self.input_db), original_length=self.original_length, sample_rate=self.sample_rate, chunk_length=self.chunk_length,
                    channels=self.channels, padding=self.padding, dac_version=SUPPORTED_VERSIONS[-1])
        path = Path(path).with_suffix(".dac")
        with open(path, "wb") as f:
            np.save(f, {"codes": self.codes.detach().cpu().numpy().astype(np.uint16), "metadata": meta})
        return path

    @classmethod
    def load(cls, path) -> "DACFile":
        blob = np.load(path, allow_pickle=True)[()]
        if blob["metadata"].get("dac_version", None) not in SUPPORTED_VERSIONS:
            raise RuntimeError(f"Given file {path} can't be loaded with this version of descript-audio-codec.")
        return cls(codes=torch.from_numpy(blob["codes"].astype(np.int32)), **blob["metadata"])


class CodecMixin:
    """``compress`` / ``decompress`` for a model with ``sample_rate``, ``hop_length``, ``preprocess``, ``encode``, ``decode`` and ``quantizer.from_codes``."""

    delay = 0   # get_delay() of the reference on this architecture (module docstring)

    @property
    def padding(self) -> bool:
        return getattr(self, "_padding", True)

    @padding.setter
    def padding(self, value: bool):
        assert isinstance(value, bool)
        self._padding = value   # (no conv of this model changes with it: module docstring)

    def get_output_length(self, input_length: int) -> int:
        return int(input_length)

    def get_delay(self) -> int:
        return 0

    def compress(self, audio_path, win_duration: Optional[float] = 1.0, normalize_db: Optional[float] = -16, n_quantizers: Optional[int] = None) -> DACFile:
        """base.py:123-196.  ``audio_path``: a file (read through ``mlx_audio_amd.audio_io.read``) or a ``(signal [n], sample_rate)`` pair."""
        if isinstance(audio_path, (tuple, list)):
            signal, sr = audio_path
        else:
            from ....audio_io import read as audio_read

            signal, sr = audio_read(audio_path)
        if sr != self.sample_rate:
            raise ValueError(f"Sample rate of the audio signal ({sr}) does not match the sample rate of the model ({self.sample_rate}).")
        arr = np.asarray(signal)
        if arr.ndim > 1 and min(arr.shape) > 1:   # (the reference fails on such input too; flattening [frames, channels] would interleave the channels into one stream)
            raise ValueError(f"compress() takes a mono signal; got an array of shape {arr.shape} -- encode each channel separately")
        x = torch.as_tensor(arr, dtype=torch.float32).reshape(-1)
        nt = x.numel()
        duration = nt / sr
        keep = self.padding
        rms = torch.sqrt((x * x).mean() + 1e-12)
        input_db = 20 * torch.log10(rms / 1.0 + 1e-12)
        if normalize_db is not None:
            x = x * torch.pow(torch.tensor(10.0), (normalize_db - input_db) / 20)
        win = duration if win_duration is None else win_duration
        if duration <= win:
            self.padding, n_samples, hop = True, nt, nt
        else:
            self.padding = False
            n_samples = int(math.ceil(int(win * self.sample_rate) / self.hop_length) * self.hop_length)
            hop = self.get_output_length(n_samples)
        pieces, chunk_length = [], 0
        for i in range(0, nt, hop):
            w = x[i:i + n_samples]
            w = torch.nn.functional.pad(w, (0, max(0, n_samples - w.numel())))
            _, c, *_ = self.encode(self.preprocess(w[None, None, :], self.sample_rate), n_quantizers)
            pieces.append(c)
            chunk_length = int(c.shape[-1])
        out = DACFile(codes=torch.cat(pieces, dim=-1), chunk_length=chunk_length, original_length=duration, input_db=float(input_db), channels=1,
                      sample_rate=int(sr), padding=self.padding, dac_version=SUPPORTED_VERSIONS[-1])
        self.padding = keep
        return out

    def decompress(self, obj: Union[str, Path, DACFile]) -> torch.Tensor:
        """base.py:198-231: -> [1, samples]."""
        if isinstance(obj, (str, Path)):
            obj = DACFile.load(obj)
        if self.sample_rate != obj.sample_rate:
            raise ValueError(f"Sample rate of the audio signal ({obj.sample_rate}) does not match the sample rate of the model ({self.sample_rate}).")
        keep = self.padding
        self.padding = bool(obj.padding)
        parts = []
        for i in range(0, obj.codes.shape[-1], obj.chunk_length):
            z = self.quantizer.from_codes(obj.codes[..., i:i + obj.chunk_length])[0]
            parts.append(self.decode(z))
        recons = torch.cat(parts, dim=1).squeeze(-1)
        recons = recons * (10.0 ** ((float(obj.input_db) - (-16)) / 20))
        self.padding = keep
        return recons
