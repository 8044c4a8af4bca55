"""KittenTTS behind the reference's model protocol (``tts/models/kitten_tts/kitten_tts.py:94-751``), computing on MI355X through ``KittenEngine``.

Same as the reference: ``ModelConfig`` fields, ``sanitize`` (dot-form Snake names of older exports), ``post_load_hook`` (``voices.npz``), the
token-id ``__call__(input_ids, ref_s, speed, return_output)`` contract with its ``Output`` record, voice aliases / speed priors, the voice-row
choice ``min(len(text), rows - 1)``, ``generate()``'s chunking, cross-fade, tail handling and ``GenerationResult`` fields, ``sample_rate``.

Host-side text processing is the caller's: the reference imports ``phonemizer`` (espeak) lazily and raises its install hint when it is missing
(kitten_tts.py:320-334) -- so does this class; and its ~1200-line number / currency / unit normaliser (``preprocess.py``, ``clean_text=True``)
is not part of the accelerated path: ``generate(clean_text=True)`` needs ``model.text_preprocessor`` to be set to a callable (the reference's
``TextPreprocessor`` instance drops in) and says so otherwise.  ``batch_call`` is the batched entry the reference lacks (it is batch-1).
"""
from __future__ import annotations

import importlib
import re
import time
from dataclasses import dataclass
from numbers import Number
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from ..base import BaseModelArgs, GenerationResult, format_duration

PHONEMIZER_INSTALL_MESSAGE = (
    "KittenTTS requires the optional 'phonemizer-fork' package for text processing. "
    "Install it with: pip install phonemizer-fork"
)


def basic_english_tokenize(text: str) -> List[str]:
    """Words and single punctuation marks (kitten_tts.py:25-29)."""
    return re.findall(r"\w+|[^\w\s]", text)


def ensure_punctuation(text: str) -> str:
    """Stripped text, with a comma appended when it does not end in punctuation (kitten_tts.py:32-39)."""
    text = text.strip()
    return text + "," if text and text[-1] not in ".!?,;:" else text


def chunk_text(text: str, max_len: int = 400) -> List[str]:
    """Sentences (split at runs of ``.!?``), over-long ones greedily re-split at word boundaries (kitten_tts.py:42-69)."""
    chunks: List[str] = []
    for sentence in (s.strip() for s in re.split(r"[.!?]+", text)):
        if not sentence:
            continue
        if len(sentence) <= max_len:
            chunks.append(ensure_punctuation(sentence))
            continue
        cur = ""
        for word in sentence.split():
            if len(cur) + len(word) + 1 <= max_len:
                cur = f"{cur} {word}" if cur else word
            else:
                if cur:
                    chunks.append(ensure_punctuation(cur.strip()))
                cur = word
        if cur:
            chunks.append(ensure_punctuation(cur.strip()))
    return chunks


class TextCleaner:
    """Symbol -> id table of the KittenTTS checkpoints (kitten_tts.py:72-91); unknown characters are dropped."""
    PAD = "$"
    PUNCTUATION = ';:,.!?¡¿—…"«»"" '
    LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
    IPA = "ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ"

    def __init__(self):
        # a dict comprehension over the concatenation: a repeated symbol keeps its LAST index, as in the reference
        self.word_index_dictionary = {s: i for i, s in enumerate([self.PAD, *self.PUNCTUATION, *self.LETTERS, *self.IPA])}

    def __call__(self, text: str) -> List[int]:
        table = self.word_index_dictionary
        return [table[ch] for ch in text if ch in table]


@dataclass
class ModelConfig(BaseModelArgs):
    hidden_dim: int
    max_conv_dim: int
    max_dur: int
    n_layer: int
    n_mels: int
    n_token: int
    style_dim: int
    text_encoder_kernel_size: int
    asr_res_dim: int
    plbert: dict
    istftnet: dict
    sample_rate: int = 24000
    decoder_out_dim: Optional[int] = None
    voices_path: str = "voices.npz"
    speed_priors: Optional[dict] = None
    voice_aliases: Optional[dict] = None
    model_path: Optional[str] = None
    activation_quant_modules: Optional[List[str]] = None


class Model:
    @dataclass
    class Output:
        audio: torch.Tensor
        pred_dur: Optional[torch.Tensor] = None

    def __init__(self, config: ModelConfig, device: str = "cuda", precision: Optional[int] = None):
        self.config = config
        self.speed_priors = config.speed_priors or {}
        self.voice_aliases = config.voice_aliases or {}
        self.context_length = int(config.plbert["max_position_embeddings"])
        self.device = device
        self.precision = precision
        self.engine = None  # built by load_weights
        self.voices: Dict[str, np.ndarray] = {}
        self.text_preprocessor: Optional[Callable[[str], str]] = None
        self._text_cleaner = TextCleaner()
        self._phonemizer = None
        self.model_path = None

    # ------------------------------------------------------------------ checkpoint handling
    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Older exports name the Snake parameters ``alpha1.0``; the modules hold ``alpha1_0`` (kitten_tts.py:301-311)."""
        if any(".alpha1." in k or ".alpha2." in k for k in weights) and not any("alpha1_" in k or "alpha2_" in k for k in weights):
            return {k.replace(".alpha1.", ".alpha1_").replace(".alpha2.", ".alpha2_"): v for k, v in weights.items()}
        return weights

    def load_weights(self, weights, strict: bool = True):
        from .engine import KittenEngine

        w = dict(weights)
        dtypes = {v.dtype for v in w.values() if v.is_floating_point()}
        pdt = torch.bfloat16 if torch.bfloat16 in dtypes else (torch.float16 if torch.float16 in dtypes else torch.float32)
        cfg = self.config if isinstance(self.config, dict) else self.config.__dict__
        try:
            self.engine = KittenEngine({k: v.to(torch.float32) for k, v in w.items()}, cfg, device=self.device, param_dtype=pdt,
                                       precision=self.precision)
        except KeyError as e:
            if strict:
                raise ValueError(f"KittenTTS checkpoint is missing parameter {e}") from e
            raise
        return self

    def eval(self):
        return self

    @classmethod
    def post_load_hook(cls, model, model_path):
        voices_path = Path(model_path) / model.config.voices_path
        if voices_path.exists():
            model._load_voices(voices_path)
        return model

    def _load_voices(self, path):
        voices = np.load(path)
        self.voices = {k: voices[k].astype(np.float32) for k in voices.files}

    @property
    def sample_rate(self) -> int:
        return self.config.sample_rate

    # ------------------------------------------------------------------ text front end (host)
    def _get_phonemizer_backend(self):
        try:
            backend = importlib.import_module("phonemizer.backend")
        except ImportError as exc:
            raise ImportError(PHONEMIZER_INSTALL_MESSAGE) from exc
        return backend.EspeakBackend

    def _get_phonemizer(self):
        if self._phonemizer is None:
            self._phonemizer = self._get_phonemizer_backend()(language="en-us", preserve_punctuation=True, with_stress=True)
        return self._phonemizer

    def _prepare_inputs(self, text: str, voice: str, speed: float, clean_text: bool):
        """text -> (token ids ``[1, T]`` with the 0 BOS / EOS, voice row ``[1, 256]``, effective speed) (kitten_tts.py:340-370)."""
        voice = self.voice_aliases.get(voice, voice)
        if voice not in self.voices:
            raise ValueError(f"Voice '{voice}' not available. Choose from: {sorted(self.voices.keys())}")
        if voice in self.speed_priors:
            speed = speed * self.speed_priors[voice]
        if clean_text:
            if self.text_preprocessor is not None:
                text = self.text_preprocessor(text)
            elif not getattr(self, "_warned_no_preprocessor", False):
                # the reference's TextPreprocessor (number / currency / unit expansion, kitten_tts.py:30-330) is host-side text normalisation outside
                # the hot path; a default generate() call must still work out of the box, so without one the text goes to the phonemizer as it is
                import warnings

                warnings.warn("KittenTTS.generate(clean_text=True) without model.text_preprocessor: the text is phonemized un-normalised "
                              "(set model.text_preprocessor to a callable for the reference's number / unit expansion)", stacklevel=3)
                self._warned_no_preprocessor = True
        phonemes = " ".join(basic_english_tokenize(self._get_phonemizer().phonemize([text])[0]))
        tokens = [0, *self._text_cleaner(phonemes), 0]
        rows = self.voices[voice]
        ref_id = min(len(text), rows.shape[0] - 1)
        return torch.tensor([tokens], dtype=torch.int32), torch.from_numpy(rows[ref_id:ref_id + 1].reshape(1, -1)), speed

    # ------------------------------------------------------------------ forward
    def _require_engine(self):
        if self.engine is None:
            raise RuntimeError("Model has no weights: call load_weights() (or mlx_audio_amd.tts.utils.load_model)")

    def __call__(self, input_ids: torch.Tensor, ref_s: torch.Tensor, speed: Number = 1.0, return_output: bool = False):
        """token ids ``[1, T]`` (BOS / EOS included) + voice row ``[1, 256]`` -> waveform ``[1, N]``... the reference returns ``decoder(...)[0]``
        of a ``[1, 1, N]`` tensor, i.e. ``[1, N]`` (kitten_tts.py:376-413)."""
        self._require_engine()
        ids = torch.as_tensor(input_ids).reshape(-1).to(torch.long)
        outs, durs = self.engine.forward([ids], torch.as_tensor(ref_s, dtype=torch.float32).reshape(1, -1), speed=float(speed))
        audio = outs[0][None, :]
        return self.Output(audio=audio, pred_dur=durs[0]) if return_output else audio

    def batch_call(self, input_ids: Sequence[torch.Tensor], ref_s: torch.Tensor, speed: Number = 1.0):
        """Ragged batch of utterances in one launch sequence: list of token-id vectors + ``[B, 256]`` voice rows -> (list of waveforms, list of
        predicted durations).  Every utterance's result equals its single ``__call__`` (the quantisation extrema are per utterance)."""
        self._require_engine()
        return self.engine.forward([torch.as_tensor(i).reshape(-1).to(torch.long) for i in input_ids],
                                   torch.as_tensor(ref_s, dtype=torch.float32).reshape(len(input_ids), -1), speed=float(speed))

    # ------------------------------------------------------------------ generate (kitten_tts.py:419-751)
    def _trim_tail_artifact(self, audio: torch.Tensor) -> torch.Tensor:
        """Cuts a spurt of sound that follows >= 30 ms of near-silence within the last second (10 ms RMS frames relative to the tail's loudest frame:
        silence < 0.1, sound > 0.2), scanning from the end for the first such silent run (kitten_tts.py:448-489)."""
        sr = self.sample_rate
        hop = max(1, int(sr * 0.01))
        tail_len = min(int(audio.shape[0]), int(sr * 1.0))
        if tail_len <= hop * 3:
            return audio
        tail = audio[-tail_len:].detach().float().cpu().numpy()
        n_frames = tail.shape[0] // hop
        if n_frames <= 3:
            return audio
        frames = tail[-n_frames * hop:].reshape(n_frames, hop)
        rms = np.sqrt(np.mean(frames * frames, axis=1))
        if rms.max() <= 1e-6:
            return audio
        rel = rms / (rms.max() + 1e-9)
        min_silence = max(3, int(0.03 / 0.01))
        run = 0
        for i in range(len(rel) - 1, -1, -1):
            if rel[i] < 0.1:
                run += 1
                continue
            if run >= min_silence:
                low_end = i + run
                if np.any(rel[low_end + 1:] > 0.2):
                    return audio[: int(audio.shape[0]) - tail_len + (low_end + 1) * hop]
                return audio
            run = 0
        return audio

    def _apply_tail(self, audio: torch.Tensor, fade_out_ms: int, tail_silence_ms: int) -> torch.Tensor:
        sr = self.sample_rate
        fade_out_samples = int(sr * max(fade_out_ms, 0) / 1000)
        tail_silence_samples = int(sr * max(tail_silence_ms, 0) / 1000)
        try:
            audio = self._trim_tail_artifact(audio)
        except Exception:  # the reference treats the trim as best-effort (kitten_tts.py:488-489)
            pass
        n = int(audio.shape[0])
        if fade_out_samples > 0:
            hop = max(1, int(sr * 0.01))
            tail_len = min(n, int(sr * max(fade_out_ms, 400) / 1000))
            fade_start = max(0, n - fade_out_samples)
            if tail_len > hop:  # fade from the last energetic 10 ms frame near the end
                tail = audio[-tail_len:].detach().float().cpu().numpy()
                n_frames = tail.shape[0] // hop
                if n_frames > 0:
                    frames = tail[-n_frames * hop:].reshape(n_frames, hop)
                    rms = np.sqrt(np.mean(frames * frames, axis=1))
                    loud = np.where(rms > max(rms.max() * 0.05, 1e-4))[0]
                    if len(loud):
                        fade_start = n - tail_len + int(loud[-1]) * hop
            fade_len = n - fade_start
            if fade_len < fade_out_samples:
                fade_start = max(0, n - fade_out_samples)
                fade_len = n - fade_start
            if fade_len > 0:
                fade_start, fade_len = int(fade_start), int(fade_len)
                curve = 1.0 - torch.arange(fade_len, dtype=audio.dtype, device=audio.device) / fade_len
                audio = torch.cat([audio[:fade_start], audio[fade_start:] * curve])
        if tail_silence_samples > 0:
            audio = torch.cat([audio, torch.zeros(tail_silence_samples, dtype=audio.dtype, device=audio.device)])
        return audio

    @staticmethod
    def _crossfade(prev: torch.Tensor, nxt: torch.Tensor, crossfade_samples: int):
        fade = min(crossfade_samples, int(prev.shape[0]), int(nxt.shape[0]))
        if fade <= 0:
            return prev, nxt
        t = torch.arange(fade, dtype=prev.dtype, device=prev.device) / fade
        return torch.cat([prev[:-fade], prev[-fade:] * (1.0 - t) + nxt[:fade] * t]), nxt[fade:]

    def _result(self, audio: torch.Tensor, segment_idx: int, token_count: int, seconds: float) -> GenerationResult:
        samples = int(audio.shape[0])
        assert samples > 0, "No audio generated"
        dur = samples / self.sample_rate
        return GenerationResult(
            audio=audio, samples=samples, sample_rate=self.sample_rate, segment_idx=segment_idx, token_count=token_count,
            audio_duration=format_duration(dur), real_time_factor=round(seconds / dur if dur > 0 else 0, 2),
            prompt={"tokens": token_count, "tokens-per-sec": round(token_count / seconds, 2) if seconds > 0 else 0},
            audio_samples={"samples": samples, "samples-per-sec": round(samples / seconds, 2) if seconds > 0 else 0},
            processing_time_seconds=seconds, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0)

    def generate(self, text: str, voice: str = "expr-voice-5-m", speed: float = 1.0, clean_text: bool = True, chunk_size: int = 400,
                 crossfade_ms: int = 20, fade_out_ms: int = 200, tail_silence_ms: int = 200, **kwargs):
        """Generator of ``GenerationResult``: chunks are synthesised one by one, each emitted once its successor exists (cross-faded over
        ``crossfade_ms``), the last one after trimming / fade-out / trailing silence.  The reference carries ``speed`` from chunk to chunk, so a
        voice's speed prior compounds per chunk (kitten_tts.py:563-566 rebinds ``speed``); reproduced."""
        if not self.voices:
            raise RuntimeError("Voices are not loaded. Ensure voices.npz is present.")
        text = text.strip()
        if not text:
            return
        chunks = [ensure_punctuation(text)] if len(text) <= chunk_size else chunk_text(text, max_len=chunk_size)
        crossfade_samples = int(self.sample_rate * max(crossfade_ms, 0) / 1000)
        t0 = time.time()
        pending, pending_tokens, emit_idx = None, 0, 0
        for chunk in chunks:
            input_ids, ref_s, speed = self._prepare_inputs(chunk, voice, speed, clean_text)
            audio = self(input_ids, ref_s, speed).reshape(-1)
            if pending is None:
                pending, pending_tokens = audio, int(input_ids.shape[-1])
                continue
            out_audio, pending = self._crossfade(pending, audio, crossfade_samples)
            tokens, pending_tokens = pending_tokens, int(input_ids.shape[-1])
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = time.time()
            seconds, t0 = now - t0, now
            yield self._result(out_audio, emit_idx, tokens, seconds)
            emit_idx += 1
        if pending is not None:
            pending = self._apply_tail(pending, fade_out_ms, tail_silence_ms)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            yield self._result(pending, emit_idx, pending_tokens, time.time() - t0)
