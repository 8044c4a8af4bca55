"""Kokoro-82M behind the reference's model protocol (``Model(config)``, ``sanitize``, ``__call__``,
``generate`` -- ``tts/models/kokoro/kokoro.py:38-367``), computing on MI355X through ``KokoroEngine``.

What is the same as the reference: constructor / config fields, checkpoint key handling (``sanitize``), the
phoneme-string ``__call__(phonemes, ref_s, speed, return_output)`` contract and its ``Output`` record, the
``generate()`` generator and every field of the ``GenerationResult`` it yields, ``sample_rate``.

What differs, on purpose (DESIGN.md "Reference quirks"):
  * ``GenerationResult.samples`` / ``samples-per-sec`` report the true sample count; the reference reports
    ``audio.shape[0]`` of a ``[1, N]`` array, i.e. 1 (kokoro.py:320,359-364).  Duration and RTF agree.
  * ``batch_generate`` exists (the reference's Kokoro is batch-1): utterances are batched per launch and, under
    ``torch.distributed``, sharded over the node's GPUs (mlx_audio_amd/shard.py).
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from numbers import Number
from typing import Dict, List, Optional, Sequence, Union

import torch

from ..base import BaseModelArgs, BatchGenerationResult, GenerationResult, check_array_shape, format_duration

_LSTM_KEYS = {  # PyTorch nn.LSTM parameter suffix -> the reference's hand-rolled LSTM names (kokoro.py:15-35)
    "weight_ih_l0_reverse": "Wx_backward", "weight_hh_l0_reverse": "Wh_backward",
    "bias_ih_l0_reverse": "bias_ih_backward", "bias_hh_l0_reverse": "bias_hh_backward",
    "weight_ih_l0": "Wx_forward", "weight_hh_l0": "Wh_forward",
    "bias_ih_l0": "bias_ih_forward", "bias_hh_l0": "bias_hh_forward",
}


def sanitize_lstm_weights(key: str, value) -> dict:
    base, _, suffix = key.rpartition(".")
    new = _LSTM_KEYS.get(suffix)
    return {f"{base}.{new}": value} if new else {key: value}


@dataclass
class ModelConfig(BaseModelArgs):
    istftnet: dict
    dim_in: int
    dropout: float
    hidden_dim: int
    max_conv_dim: int
    max_dur: int
    multispeaker: bool
    n_layer: int
    n_mels: int
    n_token: int
    style_dim: int
    text_encoder_kernel_size: int
    plbert: dict
    vocab: Dict[str, int]
    sample_rate: int = 24000


def _to_mlx_conv_layout(t: torch.Tensor) -> torch.Tensor:
    """PyTorch (out, in, K) -> MLX (out, K, in) unless the shape heuristic says it already is."""
    return t if check_array_shape(t) else t.permute(0, 2, 1).contiguous()


class Model:
    REPO_ID = "prince-canuma/Kokoro-82M"

    @dataclass
    class Output:
        audio: torch.Tensor
        pred_dur: Optional[torch.Tensor] = None

    def __init__(self, config: ModelConfig, repo_id: str = None, device: str = "cuda", precision: Optional[int] = None):
        self.repo_id = repo_id
        self.config = config
        self.vocab = config.vocab
        self.context_length = int(config.plbert["max_position_embeddings"])
        self.device = device
        self.precision = precision
        self.engine = None  # built by load_weights (needs the checkpoint)
        self._pipelines: Dict[str, "KokoroPipeline"] = {}
        self.model_path = None

    # ------------------------------------------------------------------ checkpoint handling
    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Checkpoint keys / layouts -> the parameter names the reference's modules use (kokoro.py:178-275,
        istftnet.py:998-1011): drop ``position_ids``, ``gamma/beta`` -> ``weight/bias``, PyTorch LSTM names,
        PyTorch conv layouts -> (out, K, in)."""
        packed = any(k.endswith((".scales", ".biases")) for k in weights)
        out: Dict[str, torch.Tensor] = {}
        for key, v in weights.items():
            if key.startswith("bert_encoder"):
                out[key] = v
            elif key.startswith("bert"):
                if "position_ids" not in key:
                    out[key] = v
            elif key.startswith(("text_encoder", "predictor")):
                suffix = key.rsplit(".", 1)[-1]
                if key.startswith("text_encoder") and suffix in ("gamma", "beta"):
                    out[key.rsplit(".", 1)[0] + (".weight" if suffix == "gamma" else ".bias")] = v
                elif key.startswith("predictor") and ("F0_proj.weight" in key or "N_proj.weight" in key):
                    out[key] = v if packed else v.permute(0, 2, 1).contiguous()
                elif "weight_v" in key:
                    out[key] = v if packed else _to_mlx_conv_layout(v)
                elif suffix in _LSTM_KEYS:
                    out.update(sanitize_lstm_weights(key, v))
                else:
                    out[key] = v
            elif key.startswith("decoder"):
                if packed:
                    out[key] = v
                elif "noise_convs" in key and key.endswith(".weight"):
                    out[key] = v.permute(0, 2, 1).contiguous()
                elif "weight_v" in key:
                    out[key] = _to_mlx_conv_layout(v)
                else:
                    out[key] = v
        return out

    def load_weights(self, weights, strict: bool = True):
        """``weights``: dict or list of (name, tensor) pairs with the sanitized names; builds the device engine
        (weight-norm folding, bf16 packing into MFMA fragment order, upload)."""
        from .engine import KokoroEngine

        w = dict(weights)
        if any(k.endswith((".scales", ".biases")) for k in w):
            raise ValueError("packed quantized Kokoro checkpoints are not supported by the MI355X engine (bf16 / fp32 only)")
        dtypes = {v.dtype for v in w.values() if v.is_floating_point()}
        pdt = torch.bfloat16 if torch.bfloat16 in dtypes else (torch.float16 if torch.float16 in dtypes else torch.float32)
        cfg = self.config if isinstance(self.config, dict) else self.config.__dict__
        try:
            # one rule for every entry point (KokoroEngine.default_precision): bf16 checkpoints -> 6 (the benchmarked mode), fp16 -> 2 (held exactly
            # by the bf16 hi + lo images), float32 -> 4 (fp16 weight images + fp16 hi / lo activations), unless the caller chose a mode
            prec = self.precision if self.precision is not None else KokoroEngine.default_precision(pdt)
            self.engine = KokoroEngine({k: v.to(torch.float32) for k, v in w.items()}, cfg, device=self.device,
                                       param_dtype=pdt, precision=prec)
        except KeyError as e:
            if strict:
                raise ValueError(f"Kokoro checkpoint is missing parameter {e}") from e
            raise
        return self

    def eval(self):
        return self

    @property
    def sample_rate(self) -> int:
        return self.config.sample_rate

    # ------------------------------------------------------------------ forward
    def phonemes_to_ids(self, phonemes: str) -> torch.Tensor:
        ids = [self.vocab[p] for p in phonemes if self.vocab.get(p) is not None]
        assert len(ids) + 2 <= self.context_length, (len(ids) + 2, self.context_length)
        return torch.tensor([0, *ids, 0], dtype=torch.long)

    def __call__(self, phonemes: str, ref_s: torch.Tensor, speed: Number = 1, return_output: bool = False, decoder=None):
        """phonemes -> waveform ``[1, N]`` (kokoro.py:111-177).  ``decoder`` is accepted for signature parity."""
        if self.engine is None:
            raise RuntimeError("Model has no weights: call load_weights() (or mlx_audio_amd.tts.utils.load_model)")
        ids = self.phonemes_to_ids(phonemes)
        outs, durs = self.engine.forward([ids], ref_s.reshape(1, -1), speed=float(speed))
        audio = outs[0][None, :]
        return self.Output(audio=audio, pred_dur=durs[0]) if return_output else audio

    def _get_pipeline(self, lang_code: str):
        from .pipeline import KokoroPipeline

        if lang_code not in self._pipelines:
            self._pipelines[lang_code] = KokoroPipeline(lang_code=lang_code, model=self,
                                                        repo_id=self.REPO_ID if self.repo_id is None else self.repo_id)
        return self._pipelines[lang_code]

    def _result(self, audio: torch.Tensor, segment_idx: int, token_count: int, seconds: float) -> GenerationResult:
        samples = int(audio.shape[-1])
        dur = samples / self.sample_rate
        return GenerationResult(
            audio=audio.reshape(-1), samples=samples, sample_rate=self.sample_rate, segment_idx=segment_idx,
            token_count=token_count, audio_duration=format_duration(dur),
            real_time_factor=round(seconds / dur, 2) if dur > 0 else 0,
            prompt={"tokens": token_count, "tokens-per-sec": round(token_count / seconds, 2) if seconds > 0 else 0},
            audio_samples={"samples": samples, "samples-per-sec": round(samples / seconds, 2) if seconds > 0 else 0},
            processing_time_seconds=seconds, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0)

    def generate(self, text: str, voice: str = None, speed: float = 1.0, lang_code: str = "a", split_pattern: str = r"\n+", **kwargs):
        """Generator of ``GenerationResult`` per text segment (kokoro.py:293-367); unknown kwargs are ignored as the
        reference's CLI passes many (tts/generate.py:320-346)."""
        pipeline = self._get_pipeline(lang_code)
        pipeline.voices = {}
        voice = voice or "af_heart"
        t0 = time.time()
        for segment_idx, (graphemes, phonemes, audio) in enumerate(pipeline(text, voice=voice, speed=speed, split_pattern=split_pattern)):
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = time.time()
            seconds, t0 = now - t0, now
            assert audio is not None and audio.shape[-1] > 0, "No audio generated"
            yield self._result(audio, segment_idx, len(phonemes) if phonemes is not None else 0, seconds)

    def batch_generate(self, texts: Sequence[str], voices: Union[None, str, torch.Tensor, Sequence[Union[str, torch.Tensor]]] = None, speed: float = 1.0,
                       lang_code: str = "a", split_pattern: str = r"\n+", **kwargs):
        """All utterances in one launch sequence; yields ``BatchGenerationResult`` (tts/models/base.py:88-99) with ``sequence_idx`` = input position.

        Two input forms: (a) ``voices`` are style tensors (one ``[256]`` / ``[1, 256]`` row per utterance, or a ``[B, 256]`` tensor): ``texts`` are
        PHONEME strings of at most 510 symbols and everything runs as one engine pass, results in input order; (b) ``voices`` are voice names
        (or ``None`` = "af_heart", one name for all or one per text): ``texts`` go through G2P and chunking and the chunks of all texts are
        batched step by step through ``KokoroBatchSession`` (results in completion order).  The reference's Kokoro has no batch path
        (kokoro.py:126 is batch-1); the signature is the one its serving shell probes for (server.py:519-545: ``texts`` / ``voices``)."""
        tensor_voices = isinstance(voices, torch.Tensor) or (isinstance(voices, (list, tuple)) and len(voices) > 0 and isinstance(voices[0], torch.Tensor))
        if tensor_voices:
            ids = [self.phonemes_to_ids(p) for p in texts]
            ref = voices if isinstance(voices, torch.Tensor) else torch.cat([v.reshape(1, -1) for v in voices], 0)
            t0 = time.time()
            outs, _ = self.engine.forward(ids, ref, speed=float(speed))
            torch.cuda.synchronize()
            seconds = time.time() - t0
            for i, a in enumerate(outs):
                n = int(a.numel())
                yield BatchGenerationResult(audio=a, sequence_idx=i, samples=n, sample_rate=self.sample_rate, token_count=len(texts[i]),
                                            audio_duration=format_duration(n / self.sample_rate), processing_time_seconds=seconds,
                                            peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9)
            return
        from ...continuous import TTSBatchItem, TTSBatchOptions

        names = [voices] * len(texts) if voices is None or isinstance(voices, str) else list(voices)
        if len(names) != len(texts):
            raise ValueError(f"batch_generate: {len(texts)} texts but {len(names)} voices")
        session = self.create_tts_batch_session(TTSBatchOptions(lang_code=lang_code, max_batch_size=max(1, len(texts))))
        session.add([TTSBatchItem(sequence_id=i, text=t, voice=v, speed=speed, extra={"split_pattern": split_pattern}) for i, (t, v) in enumerate(zip(texts, names))])
        t0 = time.time()
        while not session.idle:
            for ev in session.step():
                if ev.error is not None:
                    raise ev.error
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                yield BatchGenerationResult(audio=ev.audio, sequence_idx=ev.sequence_id, samples=ev.samples, sample_rate=self.sample_rate,
                                            token_count=ev.token_count, audio_duration=format_duration(ev.samples / self.sample_rate),
                                            processing_time_seconds=time.time() - t0,
                                            peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0,
                                            is_final_chunk=True)

    # ------------------------------------------------------------------ continuous batching hooks (tts/continuous.py; probed by server.py:485-600)
    def supports_tts_batch(self, *, stream: bool = False, voice: Optional[str] = None, instruct: Optional[str] = None, ref_audio=None,
                           ref_text: Optional[str] = None, speed: Optional[float] = 1.0, pitch: Optional[float] = 1.0, **kwargs) -> bool:
        """Kokoro has no instruct / reference-audio / pitch controls; any speed and chunk-level streaming batch fine."""
        del kwargs, stream, voice, speed
        return not instruct and ref_audio is None and ref_text is None and pitch in (None, 1.0)

    def supports_tts_continuous_batch(self, **kwargs) -> bool:
        return self.supports_tts_batch(**kwargs)

    def create_tts_batch_session(self, options):
        from .continuous_batching import KokoroBatchSession

        return KokoroBatchSession(self, options)
