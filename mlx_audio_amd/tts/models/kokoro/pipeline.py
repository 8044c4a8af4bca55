"""Language-aware front of Kokoro: text -> (graphemes, phonemes, audio) chunks and voice management.

Mirrors the surface of the reference's ``KokoroPipeline`` (``tts/models/kokoro/pipeline.py:91-528``): language
codes and aliases, ``load_voice`` (single names, ``a,b`` blends, ``.pt`` / ``.safetensors`` files), the 510-phoneme
chunk limit, ``Result`` records that unpack as ``(graphemes, phonemes, audio)``.  The G2P itself is the external
``misaki`` package in the reference (pipeline.py:27-59); it is not part of the hot path,
so it is a pluggable callable here: ``KokoroPipeline(..., g2p=fn)`` with ``fn(text) -> (phonemes, tokens)`` like ``misaki.en.G2P``
(tokens: objects with ``text`` / ``phonemes`` / ``whitespace``; English then takes the reference's token-level chunking, ``en_tokenize`` +
``waterfall_last``, pipeline.py:231-294, and gets word timestamps, ``join_timestamps``, :360-399) or ``fn(text) -> phoneme string`` (other
languages: the reference's 400-character text chunks, :473-528; for English a plain string is cut by ``chunk_phonemes``, an extension).  When none is given ``misaki`` is imported lazily and, if it is missing, the same
``ImportError`` + pip hint the reference's loader produces is raised at first use.  Text already in phonemes can
be synthesised without any G2P through ``generate_from_tokens``.
"""
from __future__ import annotations

import logging
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

    return g


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
        """For a G2P that returns a bare phoneme string for English (no tokens to chunk by): pieces of at most ``limit`` symbols, preferring the
        latest sentence, then clause, then word boundary before the limit (the order of ``waterfall_last``)."""
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
    def tokens_to_ps(cls, tokens) -> str:
        return "".join(t.phonemes + (" " if t.whitespace else "") for t in tokens).strip()

    @classmethod
    def tokens_to_text(cls, tokens) -> str:
        return "".join(t.text + t.whitespace for t in tokens).strip()

    @classmethod
    def waterfall_last(cls, tokens, next_count: int, waterfall=("!.?…", ":;", ",—"), bumps=(")", "”")) -> int:
        """Where to cut ``tokens`` so that what stays behind fits: after the LAST sentence mark if that leaves at most 510 phonemes for the next
        chunk, else after the last clause mark, else the last comma / dash, else everything (pipeline.py:236-261).  A closing bracket / quote that
        follows the mark goes with it."""
        for marks in waterfall:
            z = next((i for i in range(len(tokens) - 1, -1, -1) if tokens[i].phonemes in set(marks)), None)
            if z is None:
                continue
            z += 1
            if z < len(tokens) and tokens[z].phonemes in bumps:
                z += 1
            if next_count - len(cls.tokens_to_ps(tokens[:z])) <= MAX_PHONEMES:
                return z
        return len(tokens)

    def en_tokenize(self, tokens) -> Generator[tuple, None, None]:
        """English tokens -> (text, phonemes, tokens) chunks of at most 510 phonemes, cut by ``waterfall_last`` (pipeline.py:266-294).  As in the
        reference the flap is rewritten (ɾ -> T) and a missing pronunciation becomes the empty string, in place."""
        held, count = [], 0
        for t in tokens:
            t.phonemes = "" if t.phonemes is None else t.phonemes.replace("ɾ", "T")
            piece = t.phonemes + (" " if t.whitespace else "")
            ahead = count + len(piece.rstrip())
            if ahead > MAX_PHONEMES:
                z = KokoroPipeline.waterfall_last(held, ahead)
                yield KokoroPipeline.tokens_to_text(held[:z]), KokoroPipeline.tokens_to_ps(held[:z]), held[:z]
                held = held[z:]
                count = len(KokoroPipeline.tokens_to_ps(held))
                if not held:
                    piece = piece.lstrip()
            held.append(t)
            count += len(piece)
        if held:
            yield KokoroPipeline.tokens_to_text(held).strip(), KokoroPipeline.tokens_to_ps(held).strip(), held

    @classmethod
    def join_timestamps(cls, tokens, pred_dur) -> None:
        """Word start / end times from the predicted durations (pipeline.py:360-399): positions are counted in half frames (80 per second) so that the
        frame of a space can be split between its neighbours; ``pred_dur[0]`` is <bos> (all but 3 of its frames count as lead-in)."""
        if not tokens or len(pred_dur) < 3:
            return
        dur = [int(d) for d in (pred_dur.tolist() if hasattr(pred_dur, "tolist") else pred_dur)]
        left = right = 2 * max(0, dur[0] - 3)
        i = 1
        for t in tokens:
            if i >= len(dur) - 1:
                break
            if not t.phonemes:
                if t.whitespace:
                    i += 1
                    left = right + dur[i]
                    right = left + dur[i]
                    i += 1
                continue
            j = i + len(t.phonemes)
            if j >= len(dur):
                break
            t.start_ts = left / 80
            space = dur[j] if t.whitespace else 0
            left = right + 2 * sum(dur[i:j]) + space
            t.end_ts = left / 80
            right = left + space
            i = j + (1 if t.whitespace else 0)

    @staticmethod
    def text_chunks(graphemes: str, chunk_size: int = 400) -> List[str]:
        """Non-English text -> pieces of roughly ``chunk_size`` characters on sentence boundaries (pipeline.py:473-503); without any sentence mark the
        text stays one piece (the reference's character-count fallback cannot trigger: a non-empty text always yields one piece)."""
        parts = re.split(r"([.!?]+)", graphemes)
        chunks, current = [], ""
        for i in range(0, len(parts), 2):
            sentence = parts[i] + (parts[i + 1] if i + 1 < len(parts) else "")
            if len(current) + len(sentence) <= chunk_size:
                current += sentence
            else:
                if current:
                    chunks.append(current.strip())
                current = sentence
        if current:
            chunks.append(current.strip())
        if not chunks:
            chunks = [graphemes[i:i + chunk_size] for i in range(0, len(graphemes), chunk_size)]
        return chunks

    def phoneme_chunks(self, graphemes: str) -> Generator[tuple, None, None]:
        """(graphemes, phonemes, tokens or None) per synthesis call for one text segment, as ``__call__`` of the reference cuts it (pipeline.py:444-528)."""
        if self.lang_code in "ab":
            out = self.g2p(graphemes)
            tokens = out[1] if isinstance(out, tuple) and len(out) > 1 else None
            if tokens is None:   # a G2P that only returns the phoneme string
                for ps in self.chunk_phonemes((out[0] if isinstance(out, tuple) else out) or ""):
                    yield graphemes, ps, None
                return
            for gs, ps, tks in self.en_tokenize(tokens):
                if not ps:
                    continue
                if len(ps) > MAX_PHONEMES:
                    logging.warning(f"Unexpected len(ps) == {len(ps)} > {MAX_PHONEMES} and ps == '{ps}'")
                    ps = ps[:MAX_PHONEMES]
                yield gs, ps, tks
            return
        for chunk in self.text_chunks(graphemes):
            if not chunk.strip():
                continue
            out = self.g2p(chunk)
            ps = out[0] if isinstance(out, tuple) else out
            if not ps:
                continue
            if len(ps) > MAX_PHONEMES:
                logging.warning(f"Truncating len(ps) == {len(ps)} > {MAX_PHONEMES}")
                ps = ps[:MAX_PHONEMES]
            yield chunk, ps, None

    @classmethod
    def infer(cls, model, ps: str, pack: torch.Tensor, speed: Number = 1):
        return model(ps, pack[len(ps) - 1], speed, return_output=True)

    def generate_from_tokens(self, tokens: Union[str, list], voice, speed: Number = 1, model=None) -> Generator["KokoroPipeline.Result", None, None]:
        """Audio from a raw phoneme string (no G2P; ``ValueError`` beyond 510 phonemes) or from pre-processed tokens (chunked like text), as
        pipeline.py:305-358; ``ValueError`` for a missing voice."""
        model = model or self.model
        if model and voice is None:
            raise ValueError('Specify a voice: pipeline.generate_from_tokens(..., voice="af_heart")')
        pack = self.load_voice(voice) if model else None
        if isinstance(tokens, str):
            if len(tokens) > MAX_PHONEMES:
                raise ValueError(f"Phoneme string too long: {len(tokens)} > {MAX_PHONEMES}")
            yield self.Result(graphemes="", phonemes=tokens, output=KokoroPipeline.infer(model, tokens, pack, speed) if model else None)
            return
        for gs, ps, tks in self.en_tokenize(tokens):
            if not ps:
                continue
            if len(ps) > MAX_PHONEMES:
                logging.warning(f"Unexpected len(ps) == {len(ps)} > {MAX_PHONEMES} and ps == '{ps}'; truncating")
                ps = ps[:MAX_PHONEMES]
            output = KokoroPipeline.infer(model, ps, pack, speed) if model else None
            if output is not None and output.pred_dur is not None:
                KokoroPipeline.join_timestamps(tks, output.pred_dur)
            yield self.Result(graphemes=gs, phonemes=ps, tokens=tks, output=output)

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
            for gs, ps, tks in self.phoneme_chunks(graphemes):
                out = KokoroPipeline.infer(model, ps, pack, speed) if model else None
                if tks is not None and out is not None and out.pred_dur is not None:
                    KokoroPipeline.join_timestamps(tks, out.pred_dur)
                yield self.Result(graphemes=gs, phonemes=ps, tokens=tks, output=out, text_index=index)
