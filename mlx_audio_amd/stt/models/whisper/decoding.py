"""``DecodingOptions`` / ``DecodingResult`` / ``decode`` of ``mlx_audio/stt/models/whisper/decoding.py:117-165, 702-735``.

The per-token work (decoder forward, logit filters, token selection, log-prob bookkeeping: decoding.py:165-443, 588-632)
runs on the device in ``WhisperEngine.decode``; this module keeps the option / result surface and the host-side
post-processing of ``DecodingTask.run`` (slice between the first sampled token and EOT, ranking, average log-prob).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field, replace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch


def compression_ratio(text) -> float:
    text_bytes = text.encode("utf-8")
    return len(text_bytes) / len(zlib.compress(text_bytes)) if text_bytes else 0.0


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True


@dataclass(frozen=True)
class DecodingResult:
    audio_features: torch.Tensor
    language: str
    language_probs: Optional[Dict[str, float]] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


def get_suppress_tokens(tokenizer, suppress_tokens="-1") -> Tuple[int, ...]:
    """decoding.py:80-112."""
    if isinstance(suppress_tokens, str):
        suppress_tokens = [int(t) for t in suppress_tokens.split(",")]
    result = list(suppress_tokens) if suppress_tokens else []
    if -1 in result:
        result = [t for t in result if t >= 0]
        result.extend(tokenizer.non_speech_tokens)
    result.extend([tokenizer.transcribe, tokenizer.translate, tokenizer.sot, tokenizer.sot_prev, tokenizer.sot_lm])
    if tokenizer.no_speech is not None:
        result.append(tokenizer.no_speech)
    return tuple(sorted(set(result)))


def _verify_options(options: DecodingOptions) -> DecodingOptions:
    """decoding.py:512-526."""
    if options.beam_size is not None and options.best_of is not None:
        raise ValueError("beam_size and best_of can't be given together")
    if options.temperature == 0 and options.best_of is not None:
        raise ValueError("best_of with greedy sampling (T=0) is not compatible")
    if options.patience is not None and options.beam_size is None:
        raise ValueError("patience requires beam_size to be given")
    if options.length_penalty is not None and not (0 <= options.length_penalty <= 1):
        raise ValueError("length_penalty (alpha) should be a value between 0 and 1")
    if options.beam_size is not None:
        raise NotImplementedError("Beam search decoder is not yet implemented")  # decoding.py:478-479
    return options


def initial_tokens(tok, options: DecodingOptions, n_ctx: int, sample_len: int) -> List[int]:
    """decoding.py:525-551: ``[sot_prev] + prompt[-(n_ctx // 2 - 1):] + sot_sequence + prefix[-(n_ctx // 2 - sample_len):]``."""
    tokens = list(tok.sot_sequence_including_notimestamps if options.without_timestamps else tok.sot_sequence)
    if options.prefix:
        prefix = tok.encode(" " + options.prefix.strip()) if isinstance(options.prefix, str) else list(options.prefix)
        tokens = tokens + prefix[-(n_ctx // 2 - sample_len):]
    if options.prompt:
        prompt = tok.encode(" " + options.prompt.strip()) if isinstance(options.prompt, str) else list(options.prompt)
        tokens = [tok.sot_prev] + prompt[-(n_ctx // 2 - 1):] + tokens
    return [int(t) for t in tokens]


def rank_group(tokens: List[List[int]], sum_logprobs: List[float], length_penalty: Optional[float]) -> int:
    """MaximumLikelihoodRanker (decoding.py:212-235): highest sum-logprob over length (or over the Google-NMT penalty)."""
    scores = []
    for t, lp in zip(tokens, sum_logprobs):
        length = len(t)
        penalty = length if length_penalty is None else ((5 + length) / 6) ** length_penalty
        with np.errstate(divide="ignore", invalid="ignore"):
            scores.append(np.float64(lp) / penalty)
    return int(np.argmax(scores))


def decode(model, mel: torch.Tensor, options: DecodingOptions = DecodingOptions(), generator: Optional[torch.Generator] = None, **kwargs):
    """decoding.py:702-735: mel ``[n_frames, n_mels]`` or ``[*, n_frames, n_mels]`` (or already-encoded features).  ``generator`` seeds the
    device noise of ``temperature > 0`` sampling (the reference draws from MLX's global key)."""
    if single := mel.dim() == 2:
        mel = mel[None]
    if kwargs:
        options = replace(options, **kwargs)
    options = _verify_options(options)
    tok = model.get_tokenizer(language=options.language or "en", task=options.task)
    d = model.dims
    sample_len = options.sample_len or d.n_text_ctx // 2
    initial = initial_tokens(tok, options, d.n_text_ctx, sample_len)
    n_group = options.best_of or 1
    feats = mel if tuple(mel.shape[-2:]) == (d.n_audio_ctx, d.n_audio_state) else model.engine.encode(mel.to(model.engine.device))
    n_audio = feats.shape[0]
    run_feats = feats.repeat_interleave(n_group, dim=0) if n_group > 1 else feats
    suppress = get_suppress_tokens(tok, options.suppress_tokens) if options.suppress_tokens else None
    out = model.engine.decode(None, tok, sample_len=sample_len, without_timestamps=options.without_timestamps,
                              suppress_blank=options.suppress_blank, suppress_tokens=suppress,
                              max_initial_timestamp=options.max_initial_timestamp, audio_features=run_feats, initial_tokens=initial,
                              temperature=float(options.temperature), generator=generator)
    sb = out["sample_begin"]
    toks = torch.nn.functional.pad(out["tokens"], (0, 1), value=tok.eot)[:, sb:].cpu().tolist()  # GreedyDecoder.finalize
    toks = [t[: t.index(tok.eot)] for t in toks]
    sums = out["sum_logprobs"].cpu().tolist()
    nsp = out["no_speech_probs"].cpu().tolist()
    results = []
    for a in range(n_audio):
        grp = slice(a * n_group, (a + 1) * n_group)
        sel = a * n_group + (rank_group(toks[grp], sums[grp], options.length_penalty) if n_group > 1 else 0)
        t = toks[sel]
        text = tok.decode(t).strip()
        results.append(DecodingResult(audio_features=feats[a], language=options.language or "en", tokens=t, text=text,
                                      avg_logprob=sums[sel] / (len(t) + 1), no_speech_prob=nsp[a * n_group], temperature=options.temperature,
                                      compression_ratio=compression_ratio(text)))
    return results[0] if single else results
