"""Special-token view of the Whisper tokenizer (``mlx_audio/stt/models/whisper/tokenizer.py``).

The decode loop only needs the special ids; text <-> ids needs the tiktoken vocabulary files, which cannot be fetched
here (no network).  ``Tokenizer`` therefore carries the ids of the multilingual / English-only vocabularies and takes an
optional ``codec`` object with ``encode`` / ``decode`` (a tiktoken ``Encoding`` or a HF tokenizer) when one is available;
without it ``decode`` renders ids as ``<|id|>`` strings so that the pipeline stays inspectable.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

LANGUAGES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy sk te fa "
             "lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn "
             "mt sa lb my bo tl mg as tt haw ln ha ba jw su yue").split()


@dataclass
class Tokenizer:
    multilingual: bool = True
    num_languages: int = 99
    language: Optional[str] = "en"
    task: Optional[str] = "transcribe"
    codec: object = None
    blank_ids: Tuple[int, ...] = (220,)
    non_speech_tokens: Tuple[int, ...] = ()
    eot: int = field(init=False)

    def __post_init__(self):
        base = 50257 if self.multilingual else 50256
        self.eot = base
        self.sot = base + 1
        n = self.num_languages
        self.translate = self.sot + 1 + n
        self.transcribe = self.translate + 1
        self.sot_lm = self.transcribe + 1
        self.sot_prev = self.sot_lm + 1
        self.no_speech = self.sot_prev + 1
        self.no_timestamps = self.no_speech + 1
        self.timestamp_begin = self.no_timestamps + 1

    @property
    def language_token(self) -> int:
        lang = self.language or "en"
        if lang not in LANGUAGES[: self.num_languages]:
            raise KeyError(f"Language {lang} not found in tokenizer.")
        return self.sot + 1 + LANGUAGES.index(lang)

    @property
    def all_language_tokens(self) -> Tuple[int, ...]:
        return tuple(self.sot + 1 + i for i in range(self.num_languages))

    @property
    def all_language_codes(self) -> Tuple[str, ...]:
        return tuple(LANGUAGES[: self.num_languages])

    @property
    def sot_sequence(self) -> Tuple[int, ...]:
        seq = [self.sot]
        if self.multilingual:
            seq.append(self.language_token)
            seq.append(self.transcribe if self.task != "translate" else self.translate)
        return tuple(seq)

    @property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return self.sot_sequence + (self.no_timestamps,)

    def encode(self, text: str):
        if self.codec is None:
            raise RuntimeError("no vocabulary available: pass codec= (tiktoken Encoding / HF tokenizer) to Tokenizer")
        return list(self.codec.encode(text))

    def decode(self, tokens) -> str:
        toks = [int(t) for t in tokens if int(t) < self.timestamp_begin]
        if self.codec is not None:
            return self.codec.decode(toks)
        return "".join(f"<|{t}|>" for t in toks)


def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language: Optional[str] = None, task: Optional[str] = None,
                  codec=None) -> Tokenizer:
    if multilingual:
        language = language or "en"
        task = task or "transcribe"
    else:
        language, task = None, None
    return Tokenizer(multilingual=multilingual, num_languages=num_languages, language=language, task=task, codec=codec)
