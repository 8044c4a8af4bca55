"""Language-aware front of Kokoro: text -> (graphemes, phonemes, audio) chunks and voice management.

Mirrors the surface of the reference's ``KokoroPipeline`` (``tts/models/kokoro/pipeline.py:91-528``): language
codes and aliases, ``load_voice`` (single names, ``a,b`` blends, ``.pt`` / ``.safetensors`` files), the 510-phoneme
chunk limit, ``Result`` records that unpack as ``(graphemes, phonemes, audio)``.  The G2P itself is the external
``misaki`` package in the reference (pipeline.py:27-59); it is not part of the hot path,
so it is a pluggable callable here: ``KokoroPipeline(..., g2p=fn)`` with ``fn(text) -> phoneme string`` (or
``(phonemes, tokens)``).  When none is given ``misaki`` is imported lazily and, if it is missing, the same
``ImportError`` + pip hint the reference's loader produces is raised at first use.  Text already in phonemes can
be synthesised without any G2P through ``generate_from_tokens``.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass
from numbers import Number
from typing import Callable, Dict, Generator, List, Optional, Union

import torch

from .voice import load_voice_tensor

ALIASES = {"en": "a", "en-us": "a", "en-gb": "b", "es": "e", "fr-fr": "f", "fr": "f", "hi": "h", "it": "i", "pt-br": "p",
           "pt": "p", "ja": "j", "zh": "z"}
LANG_CODES = dict(a="American English", b="British English", e="es", f="fr-fr", h="hi", i="it", p="pt-br", j="Japanese",
                  z="Mandarin Chinese")
MAX_PHONEMES = 510  # context 512 minus BOS / EOS


def _default_g2p(lang_code: str) -> Callable:
    try:
        if lang_code in "ab":
            from misaki import en

            g = en.G2P(trf=False, british=lang_code == "b", fallback=None, unk="")
        elif lang_code == "j":
            from misaki import ja

            g = ja.JAG2P()
        elif lang_code == "z":
            from misaki import zh

            g = zh.ZHG2P()
        else:
            from misaki import espeak

            g = espeak.EspeakG2P(language=LANG_CODES[lang_code])
    except ImportError as e:  # same wording as the reference's loader (utils.py:305-312)
        raise ImportError(f"\nMissing dependency while loading kokoro: {e}\nPlease install it using: pip install {e.name}") from e

    def run(text):
        out = g(text)
        return out[0] if isinstance(out, tuple) else out

    return run


class KokoroPipeline:
    @dataclass
    class Result:
        graphemes: str
        phonemes: str
        tokens: Optional[list] = None
        output: Optional[object] = None
        text_index: Optional[int] = None

        @property
        def audio(self) -> Optional[torch.Tensor]:
            return None if self.output is None else self.output.audio

        @property
        def pred_dur(self):
            return None if self.output is None else self.output.pred_dur

        def __iter__(self):  # backward-compatible unpacking: graphemes, phonemes, audio
            yield self.graphemes
            yield self.phonemes
            yield self.audio

        def __getitem__(self, i):
            return [self.graphemes, self.phonemes, self.audio][i]

        def __len__(self):
            return 3

    def __init__(self, lang_code: str, model, repo_id: str, trf: bool = False, g2p: Optional[Callable] = None):
        lang_code = ALIASES.get(lang_code.lower(), lang_code.lower())
        assert lang_code in LANG_CODES, (lang_code, LANG_CODES)
        if repo_id is None:
            raise ValueError("repo_id is required to load voices")
        self.lang_code, self.repo_id, self.model = lang_code, repo_id, model
        self.voices: Dict[str, torch.Tensor] = {}
        self._g2p = g2p

    @property
    def g2p(self) -> Callable:
        if self._g2p is None:
            self._g2p = _default_g2p(self.lang_code)
        return self._g2p

    # ------------------------------------------------------------------ voices
    def _voice_path(self, voice: str) -> str:
        if voice.endswith((".pt", ".safetensors")) and os.path.exists(voice):
            return voice
        roots = [getattr(self.model, "model_path", None), self.repo_id]
        for root in roots:
            if root and os.path.isdir(str(root)):
                for ext in (".safetensors", ".pt"):
                    p = os.path.join(str(root), "voices", voice + ext)
                    if os.path.exists(p):
                        return p
        try:
            from huggingface_hub import hf_hub_download

            return hf_hub_download(repo_id=self.repo_id, filename=f"voices/{voice}.safetensors")
        except Exception as e:
            raise FileNotFoundError(f"voice {voice!r} not found locally and could not be fetched from {self.repo_id}: {e}") from e

    def load_single_voice(self, voice: str) -> torch.Tensor:
        if voice not in self.voices:
            self.voices[voice] = load_voice_tensor(self._voice_path(voice)).to(torch.float32)
        return self.voices[voice]

    def load_voice(self, voice: Union[str, torch.Tensor], delimiter: str = ",") -> torch.Tensor:
        """A voice name, a blend ``"a,b"`` (mean of the packs, pipeline.py:190-206) or a ready tensor."""
        if isinstance(voice, torch.Tensor):
            return voice
        if voice in self.voices:
            return self.voices[voice]
        packs = [self.load_single_voice(v.strip()) for v in voice.split(delimiter)]
        self.voices[voice] = packs[0] if len(packs) == 1 else torch.stack(packs).mean(0)
        return self.voices[voice]

    # ------------------------------------------------------------------ chunking + synthesis
    @staticmethod
    def chunk_phonemes(ps: str, limit: int = MAX_PHONEMES) -> List[str]:
        """Splits a phoneme string into pieces of at most ``limit`` symbols, preferring the latest sentence, then
        clause, then word boundary before the limit (the reference's ``waterfall_last`` order, pipeline.py:236-262)."""
        out = []
        ps = ps.strip()
        while len(ps) > limit:
            head = ps[:limit]
            cut = -1
            for marks in ("!.?…", ":;", ",—", " "):
                cut = max(head.rfind(m) for m in marks)
                if cut > 0:
                    break
            cut = cut + 1 if cut > 0 else limit
            out.append(ps[:cut].strip())
            ps = ps[cut:].strip()
        if ps:
            out.append(ps)
        return out

    @classmethod
    def infer(cls, model, ps: str, pack: torch.Tensor, speed: Number = 1):
        return model(ps, pack[len(ps) - 1], speed, return_output=True)

    def generate_from_tokens(self, tokens: Union[str, list], voice, speed: Number = 1, model=None) -> Generator["KokoroPipeline.Result", None, None]:
        """Audio from a raw phoneme string (no G2P).  Raises ``ValueError`` for a missing voice or > 510 phonemes,
        like the reference (pipeline.py:348-373)."""
        model = model or self.model
        if model and voice is None:
            raise ValueError('Specify a voice: pipeline.generate_from_tokens(..., voice="af_heart")')
        pack = self.load_voice(voice) if model else None
        if not isinstance(tokens, str):
            tokens = "".join(getattr(t, "phonemes", "") + (" " if getattr(t, "whitespace", "") else "") for t in tokens).strip()
        if len(tokens) > MAX_PHONEMES:
            raise ValueError(f"Phoneme string too long: {len(tokens)} > {MAX_PHONEMES}")
        yield self.Result(graphemes="", phonemes=tokens, output=KokoroPipeline.infer(model, tokens, pack, speed) if model else None)

    def __call__(self, text: Union[str, List[str]], voice=None, speed: Number = 1, split_pattern: Optional[str] = r"\n+", model=None):
        model = model or self.model
        if model and voice is None:
            raise ValueError('Specify a voice: en_us_pipeline(text="Hello world!", voice="af_heart")')
        pack = self.load_voice(voice) if model else None
        if isinstance(text, str):
            text = re.split(split_pattern, text.strip()) if split_pattern else [text]
        for index, graphemes in enumerate(text):
            if not graphemes.strip():
                continue
            for ps in self.chunk_phonemes(self.g2p(graphemes) or ""):
                out = KokoroPipeline.infer(model, ps, pack, speed) if model else None
                yield self.Result(graphemes=graphemes, phonemes=ps, output=out, text_index=index)
