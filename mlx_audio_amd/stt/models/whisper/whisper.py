"""Whisper behind the reference's model protocol (``Model(dims, dtype)``, ``sanitize``, ``embed_audio``, ``logits``,
``__call__``, ``decode``, ``generate`` -> ``STTOutput`` -- ``stt/models/whisper/whisper.py:501-1320``), computing on
MI355X through ``WhisperEngine``.

Same as the reference: constructor / config (``ModelDimensions`` incl. the HuggingFace spelling), checkpoint key handling
(``sanitize``: HF names, conv layouts, fp16 cast), ``embed_audio`` / ``logits`` / ``__call__``, ``decode(mel, options)``
and its ``DecodingResult``, ``generate(audio, *, language, task, temperature, ..., **decode_options)`` returning
``STTOutput(text, segments, language, ...)`` with 30 s windows advanced by the decoded timestamps.

``generate`` follows the reference's window loop: each 30 s window holds only its own frames (zero-padded in the log-mel domain),
temperature fallback over the ``temperature`` tuple (sampling = arg-max of ``logits / T`` + device Gumbel noise in the decode-rules kernel,
``best_of`` groups ranked like ``MaximumLikelihoodRanker``), prompt conditioning on the previous windows / ``initial_prompt`` / ``hotwords``,
``clip_timestamps``, segment cutting at consecutive timestamps.

Deliberately not carried over (host front/back ends): resampling of non-16 kHz files, word-level timestamps and
``hallucination_silence_threshold`` (DTW over cross-attention weights), streaming (AlignAtt) -- these raise ``NotImplementedError``.
``detect_language`` returns the reference's (language tokens, probability dicts) pair.
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from ..base import STTOutput
from .audio import FRAMES_PER_SECOND, HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, log_mel_spectrogram, pad_or_trim
from .decoding import DecodingOptions, DecodingResult
from .decoding import decode as decode_function
from .synthetic import ModelDimensions
from .tokenizer import LANGUAGES, get_tokenizer

ModelConfig = ModelDimensions  # alias used by load_model (whisper.py:325)

_DECODING_OPTION_NAMES = frozenset(DecodingOptions.__dataclass_fields__)

_HF_KEY_MAP = [  # whisper.py:562-585; order matters
    ("encoder.embed_positions.weight", None),
    ("decoder.embed_positions.weight", "decoder.positional_embedding"),
    ("encoder.layer_norm.", "encoder.ln_post."),
    ("decoder.layer_norm.", "decoder.ln."),
    ("encoder.layers.", "encoder.blocks."),
    ("decoder.layers.", "decoder.blocks."),
    (".self_attn_layer_norm.", ".attn_ln."),
    (".final_layer_norm.", ".mlp_ln."),
    (".encoder_attn_layer_norm.", ".cross_attn_ln."),
    (".fc1.", ".mlp1."),
    (".fc2.", ".mlp2."),
    (".self_attn.q_proj.", ".attn.query."),
    (".self_attn.k_proj.", ".attn.key."),
    (".self_attn.v_proj.", ".attn.value."),
    (".self_attn.out_proj.", ".attn.out."),
    (".encoder_attn.q_proj.", ".cross_attn.query."),
    (".encoder_attn.k_proj.", ".cross_attn.key."),
    (".encoder_attn.v_proj.", ".cross_attn.value."),
    (".encoder_attn.out_proj.", ".cross_attn.out."),
    ("decoder.embed_tokens.", "decoder.token_embedding."),
]


def _filter_decode_options(decode_options: dict) -> dict:
    return {k: v for k, v in decode_options.items() if k in _DECODING_OPTION_NAMES}


class Model:
    def __init__(self, dims: ModelDimensions, dtype: torch.dtype = torch.float16, device: str = "cuda", precision: int = 4):
        self.dims = dims
        self.dtype = dtype
        self.device = device
        self.precision = precision
        self.engine = None
        self.model_path = None
        self.codec = None  # optional vocabulary object (encode / decode)

    # ------------------------------------------------------------------ checkpoint handling
    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        is_hf = any(k.startswith("model.") for k in weights)
        out: Dict[str, torch.Tensor] = {}
        for k, v in weights.items():
            if is_hf:
                if k.startswith("model."):
                    k = k[6:]
                skip = False
                for old, new in _HF_KEY_MAP:
                    if old in k:
                        if new is None:
                            skip = True
                            break
                        k = k.replace(old, new)
                if skip:
                    continue
                if ("conv1.weight" in k or "conv2.weight" in k) and v.dim() == 3:
                    v = v.permute(0, 2, 1).contiguous()
            if v.is_floating_point() and v.dtype != self.dtype:
                v = v.to(self.dtype)
            out[k] = v
        return out

    def load_weights(self, weights, strict: bool = True):
        from .engine import WhisperEngine

        w = dict(weights)
        try:
            # the K / V caches live in the checkpoint's floating dtype like the reference's (whisper.py:360-361): fp16 -> fp16, bf16 -> bf16, fp32 -> fp32
            fdt = [v.dtype for v in w.values() if v.is_floating_point()]
            kv_dtype = max(set(fdt), key=fdt.count) if fdt else torch.float16
            if kv_dtype not in (torch.float16, torch.bfloat16, torch.float32):
                kv_dtype = torch.float32
            self.engine = WhisperEngine({k: v.to(torch.float32) for k, v in w.items() if v.is_floating_point()}, self.dims,
                                        device=self.device, precision=self.precision, kv_dtype=kv_dtype)
        except KeyError as e:
            raise ValueError(f"Whisper checkpoint is missing parameter {e}") from e
        return self

    def eval(self):
        return self

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("Model has no weights: call load_weights() (or mlx_audio_amd.stt.utils.load_model)")
        return self.engine

    # ------------------------------------------------------------------ forward surface (whisper.py:620-642)
    def embed_audio(self, mel: torch.Tensor) -> torch.Tensor:
        single = mel.dim() == 2
        out = self._need_engine().encode(mel[None] if single else mel)
        return out[0] if single else out

    def logits(self, tokens: torch.Tensor, audio_features: torch.Tensor) -> torch.Tensor:
        eng = self._need_engine()
        xa = audio_features.to(eng.device, torch.float32)
        st = eng.new_state(xa)
        hid = eng.decoder_step(tokens.to(eng.device, torch.int32).contiguous(), st)
        return eng.logits(hid)[:, :, : self.dims.n_vocab]

    def __call__(self, mel: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        return self.logits(tokens, self.embed_audio(mel))

    @property
    def is_multilingual(self) -> bool:
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self) -> int:
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def get_tokenizer(self, language: str = None, task: str = "transcribe"):
        return get_tokenizer(self.is_multilingual, num_languages=self.num_languages, language=language, task=task, codec=self.codec)

    def decode(self, mel: torch.Tensor, options: DecodingOptions = DecodingOptions(), **kwargs):
        return decode_function(self, mel, options, **kwargs)

    def detect_language(self, mel: torch.Tensor, tokenizer=None):
        """``decoding.py:20-77``: one decoder pass on <|startoftranscript|>, every token but the language tokens masked out.  Returns
        ``(language_tokens, language_probs)``: the most probable language token per clip (a 0-d tensor for a single [frames, mels] input) and one
        ``{language code: probability}`` dict per clip (a single dict for a single input), like the reference."""
        tokenizer = tokenizer or self.get_tokenizer()
        if tokenizer.language is None or tokenizer.language_token not in tokenizer.sot_sequence:
            raise ValueError("This model doesn't have language tokens so it can't perform lang id")
        single = mel.dim() == 2
        if single:
            mel = mel[None]
        if tuple(mel.shape[-2:]) != (self.dims.n_audio_ctx, self.dims.n_audio_state):
            mel = self.embed_audio(mel)
        x = torch.full((mel.shape[0], 1), tokenizer.sot, dtype=torch.int32)
        lg = self.logits(x, mel)[:, 0].float()
        ids = torch.tensor(tokenizer.all_language_tokens, device=lg.device)
        lang_logits = lg[:, ids]                                   # the mask of the reference leaves exactly these columns
        tokens = ids[lang_logits.argmax(dim=-1)]
        probs = torch.softmax(lang_logits, dim=-1).cpu()
        dicts = [{c: float(probs[i, j]) for j, c in enumerate(tokenizer.all_language_codes)} for i in range(probs.shape[0])]
        return (tokens[0], dicts[0]) if single else (tokens, dicts)

    # ------------------------------------------------------------------ generate (whisper.py:799-1320, condensed)
    def _prepare_audio(self, audio, padding: int = N_SAMPLES) -> Tuple[torch.Tensor, int]:
        if isinstance(audio, (str, Path)):   # whisper.py:826-829: a path is loaded as a mono 16 kHz waveform (WAV / PCM in this build)
            from ...utils import load_audio

            audio = load_audio(audio)
        if not isinstance(audio, torch.Tensor):
            audio = torch.from_numpy(np.asarray(audio, dtype=np.float32))
        mel = log_mel_spectrogram(audio, n_mels=self.dims.n_mels, padding=padding)
        content_frames = mel.shape[-2] - N_FRAMES if padding else mel.shape[-2]
        return mel, content_frames

    def generate(self, audio, *, verbose: Optional[bool] = None, language: Optional[str] = None, task: str = "transcribe",
                 temperature: Union[float, Tuple[float, ...]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
                 compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
                 no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
                 initial_prompt: Optional[str] = None, return_timestamps: bool = True, word_timestamps: bool = False,
                 clip_timestamps: Union[str, List[float]] = "0", hallucination_silence_threshold: Optional[float] = None,
                 hotwords: Optional[List[str]] = None, stream: bool = False, generator: Optional[torch.Generator] = None,
                 **decode_options) -> STTOutput:
        """whisper.py:799-1320.  Windows, temperature fallback, prompt conditioning and segment cutting follow the reference; the options that
        need word-level alignment (``word_timestamps``, ``hallucination_silence_threshold``) and ``stream`` raise instead of being ignored."""
        if word_timestamps or hallucination_silence_threshold is not None:
            raise NotImplementedError("word_timestamps / hallucination_silence_threshold (DTW over cross-attention) are outside the MI355X hot path")
        if stream:
            raise NotImplementedError("generate(stream=True) is outside the MI355X hot path")
        t_start = time.time()
        if hotwords:  # stt/utils.py:15-34: the vocabulary list is folded into the prompt
            terms = ", ".join(str(t).strip() for t in hotwords if t is not None and str(t).strip())
            if terms:
                initial_prompt = f"{initial_prompt}\n{terms}" if initial_prompt else terms
        decode_options = _filter_decode_options(decode_options)
        decode_options["without_timestamps"] = not return_timestamps
        mel, content_frames = self._prepare_audio(audio)
        if language is None:
            if not self.is_multilingual:
                language = "en"
            else:
                tok0 = self.get_tokenizer()
                _, probs = self.detect_language(pad_or_trim(mel, N_FRAMES, axis=-2), tok0)   # whisper.py:897-905
                language = max(probs, key=probs.get)
        decode_options.update(language=language, task=task)
        tokenizer = self.get_tokenizer(language=language, task=task)

        if isinstance(clip_timestamps, str):  # whisper.py:937-950
            clip_timestamps = [float(ts) for ts in (clip_timestamps.split(",") if clip_timestamps else [])]
        seek_points = [round(ts * FRAMES_PER_SECOND) for ts in clip_timestamps]
        if len(seek_points) == 0:
            seek_points.append(0)
        if len(seek_points) % 2 == 1:
            seek_points.append(content_frames)
        else:
            seek_points[-1] = min(content_frames, seek_points[-1])
        seek_clips = list(zip(seek_points[::2], seek_points[1::2]))

        def decode_with_fallback(segment: torch.Tensor) -> DecodingResult:
            """whisper.py:957-996: retry at the next temperature while the text is too repetitive or too unlikely, unless it is silence."""
            temperatures = [temperature] if isinstance(temperature, (int, float)) else list(temperature)
            result = None
            for t in temperatures:
                kw = dict(decode_options)
                if t > 0:
                    kw.pop("beam_size", None)
                    kw.pop("patience", None)
                else:
                    kw.pop("best_of", None)
                result = self.decode(segment, DecodingOptions(**kw, temperature=float(t)), generator=generator)
                needs_fallback = False
                if compression_ratio_threshold is not None and result.compression_ratio > compression_ratio_threshold:
                    needs_fallback = True
                if logprob_threshold is not None and result.avg_logprob < logprob_threshold:
                    needs_fallback = True
                if no_speech_threshold is not None and result.no_speech_prob > no_speech_threshold:
                    needs_fallback = False
                if not needs_fallback:
                    break
            return result

        input_stride = N_FRAMES // self.dims.n_audio_ctx
        time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE
        all_tokens: List[int] = []
        segments: List[dict] = []
        prompt_reset_since = 0
        initial_prompt_tokens: List[int] = []
        if initial_prompt is not None:
            initial_prompt_tokens = list(tokenizer.encode(" " + initial_prompt.strip()))
            all_tokens.extend(initial_prompt_tokens)
        seek = seek_clips[0][0]
        for _, seek_clip_end in seek_clips:
            while seek < seek_clip_end:
                time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
                segment_size = min(N_FRAMES, content_frames - seek, seek_clip_end - seek)
                # whisper.py:1046-1050: only THIS window's frames, padded with 0.0 in the log-mel domain (not the audio that follows)
                seg = pad_or_trim(mel[seek:seek + segment_size], N_FRAMES, axis=-2)
                decode_options["prompt"] = all_tokens[prompt_reset_since:]
                result: DecodingResult = decode_with_fallback(seg)
                tokens = list(result.tokens)
                if no_speech_threshold is not None:
                    should_skip = result.no_speech_prob > no_speech_threshold
                    if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                        should_skip = False
                    if should_skip:
                        seek += segment_size  # silent window (whisper.py:1056-1069)
                        continue
                current: List[dict] = []
                ts = [i for i, t in enumerate(tokens) if t >= tokenizer.timestamp_begin]
                consecutive = [i for i in range(1, len(tokens))
                               if tokens[i] >= tokenizer.timestamp_begin and tokens[i - 1] >= tokenizer.timestamp_begin]
                single_ending = len(tokens) >= 2 and tokens[-2] < tokenizer.timestamp_begin <= tokens[-1]

                def seg_dict(start, end, toks, _seek=seek, _result=result):
                    return dict(seek=_seek, start=float(start), end=float(end), text=tokenizer.decode([t for t in toks if t < tokenizer.eot]),
                                tokens=list(toks), temperature=_result.temperature, avg_logprob=_result.avg_logprob,
                                compression_ratio=_result.compression_ratio, no_speech_prob=_result.no_speech_prob)

                if consecutive:  # whisper.py:1144-1170: cut at consecutive timestamp pairs
                    slices = consecutive + ([len(tokens)] if single_ending else [])
                    last = 0
                    for cur in slices:
                        sl = tokens[last:cur]
                        s_pos = sl[0] - tokenizer.timestamp_begin
                        e_pos = sl[-1] - tokenizer.timestamp_begin
                        current.append(seg_dict(time_offset + s_pos * time_precision, time_offset + e_pos * time_precision, sl))
                        last = cur
                    if single_ending:
                        seek += segment_size
                    else:
                        seek += (tokens[last - 1] - tokenizer.timestamp_begin) * input_stride
                else:
                    duration = segment_size * HOP_LENGTH / SAMPLE_RATE
                    if ts and tokens[ts[-1]] != tokenizer.timestamp_begin:
                        duration = (tokens[ts[-1]] - tokenizer.timestamp_begin) * time_precision
                    current.append(seg_dict(time_offset, time_offset + duration, tokens))
                    seek += segment_size
                if verbose:
                    for sgm in current:
                        print(f"[{sgm['start']:.3f} --> {sgm['end']:.3f}] {sgm['text']}")
                for sgm in current:  # whisper.py:1269-1277: instantaneous or empty segments are cleared
                    if sgm["start"] == sgm["end"] or sgm["text"].strip() == "":
                        sgm["text"] = ""
                        sgm["tokens"] = []
                        sgm["words"] = []
                segments.extend({"id": i, **sgm} for i, sgm in enumerate(current, start=len(segments)))
                all_tokens.extend(t for sgm in current for t in sgm["tokens"])
                if not condition_on_previous_text or result.temperature > 0.5:
                    prompt_reset_since = len(all_tokens)  # whisper.py:1298-1300
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        total = time.time() - t_start
        out_tokens = all_tokens[len(initial_prompt_tokens):]
        text = tokenizer.decode([t for t in out_tokens if t < tokenizer.eot])
        n_gen = len(out_tokens)
        return STTOutput(text=text.strip(), segments=segments, language=language, generation_tokens=n_gen, total_tokens=len(all_tokens),
                         generation_tps=n_gen / total if total > 0 else 0.0, total_time=total)
