"""Continuous batching for Kokoro: a ``TTSBatchSession`` (``mlx_audio/tts/continuous.py:49-60``) over the batched MI355X engine.

The reference's Kokoro is strictly batch-1 (``input_ids = mx.array([[0, *ids, 0]])``, kokoro.py:126): its serving shell can only run one chunk at
a time for this model.  Here every ``step()`` takes the next phoneme chunk (<= 510 symbols, pipeline.py:266-293) of every active sequence and
synthesises them in ONE engine pass (ragged batch: one launch sequence for all of them), so new requests join between steps and a long text
does not block short ones.  Events follow the protocol of the reference's sessions (``qwen3_tts/continuous_batching.py``): one
``TTSBatchEvent`` per finished sequence (``done=True``, the concatenated waveform), or one per chunk when ``options.stream`` is set
(``is_streaming_chunk`` / ``is_final_chunk``); a request that fails (unknown voice, empty text, G2P error) yields an event carrying ``error``
instead of poisoning the batch.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from ...continuous import TTSBatchEvent, TTSBatchItem, TTSBatchOptions


@dataclass
class _Seq:
    sequence_id: int
    chunks: List[str]
    pack: Optional[torch.Tensor]
    speed: float
    next: int = 0
    audio: List[torch.Tensor] = field(default_factory=list)
    tokens: int = 0
    error: Optional[BaseException] = None


class KokoroBatchSession:
    def __init__(self, model, options: TTSBatchOptions):
        self.model = model
        self.options = options
        self._seqs: "OrderedDict[int, _Seq]" = OrderedDict()

    # ------------------------------------------------------------------ protocol
    @property
    def idle(self) -> bool:
        return not self._seqs

    @property
    def available_slots(self) -> int:
        return max(0, int(self.options.max_batch_size) - len(self._seqs))

    def add(self, items: List[TTSBatchItem]) -> None:
        if len(items) > self.available_slots:
            raise ValueError(f"add: {len(items)} items for {self.available_slots} free slots (max_batch_size {self.options.max_batch_size})")
        for item in items:
            if item.sequence_id in self._seqs:
                raise ValueError(f"add: sequence_id {item.sequence_id} is already active")
            self._seqs[item.sequence_id] = self._prepare(item)

    def cancel(self, sequence_id: int) -> None:
        self._seqs.pop(sequence_id, None)

    def step(self) -> List[TTSBatchEvent]:
        events: List[TTSBatchEvent] = []
        sr = self.model.sample_rate
        # failed / empty requests finish first
        for sid in [s.sequence_id for s in self._seqs.values() if s.error is not None or not s.chunks]:
            s = self._seqs.pop(sid)
            events.append(TTSBatchEvent(sequence_id=sid, done=True, sample_rate=sr, error=s.error or ValueError("no phonemes to synthesise")))
        if not self._seqs:
            return events
        # one chunk per active sequence, grouped by speed (the engine takes one speed per pass): the oldest sequence picks the group
        lead = next(iter(self._seqs.values())).speed
        batch = [s for s in self._seqs.values() if s.speed == lead]
        ids, refs = [], []
        for s in batch:
            ps = s.chunks[s.next]
            ids.append(self.model.phonemes_to_ids(ps))
            refs.append(s.pack[len(ps) - 1].reshape(1, -1))  # the voice row of this chunk length (pipeline.py:303)
        try:
            outs, _ = self.model.engine.forward(ids, torch.cat(refs, 0), speed=float(lead))
        except BaseException as e:  # the whole pass failed: every sequence in it reports the error
            for s in batch:
                self._seqs.pop(s.sequence_id, None)
                events.append(TTSBatchEvent(sequence_id=s.sequence_id, done=True, sample_rate=sr, error=e))
            return events
        for s, ps, a in zip(batch, [b.chunks[b.next] for b in batch], outs):
            s.next += 1
            s.tokens += len(ps)
            last = s.next == len(s.chunks)
            a = a.reshape(-1)
            if self.options.stream:
                events.append(TTSBatchEvent(sequence_id=s.sequence_id, audio=a, sample_rate=sr, samples=int(a.numel()), token_count=len(ps), done=last,
                                            is_streaming_chunk=True, is_final_chunk=last, metadata={"chunk_index": s.next - 1, "chunks": len(s.chunks)}))
            else:
                s.audio.append(a)
                if last:
                    full = s.audio[0] if len(s.audio) == 1 else torch.cat(s.audio)
                    events.append(TTSBatchEvent(sequence_id=s.sequence_id, audio=full, sample_rate=sr, samples=int(full.numel()), token_count=s.tokens, done=True,
                                                is_final_chunk=True, metadata={"chunks": len(s.chunks)}))
            if last:
                self._seqs.pop(s.sequence_id, None)
        return events

    # ------------------------------------------------------------------ request preparation (host side: G2P, chunking, voice pack)
    def _prepare(self, item: TTSBatchItem) -> _Seq:
        try:
            lang = item.extra.get("lang_code") or (self.options.lang_code if self.options.lang_code not in (None, "auto") else "a")
            pipeline = self.model._get_pipeline(lang)
            pack = pipeline.load_voice(item.voice or "af_heart")
            parts = re.split(item.extra.get("split_pattern", r"\n+"), item.text.strip()) if item.text else []
            chunks = [ps for g in parts if g.strip() for _, ps, _ in pipeline.phoneme_chunks(g)]
            # symbols outside the vocabulary are dropped by phonemes_to_ids (kokoro.py:123-125); a chunk left empty is skipped
            chunks = [ps for ps in chunks if any(self.model.vocab.get(p) is not None for p in ps)]
            return _Seq(item.sequence_id, chunks, pack, float(item.speed) if item.speed else 1.0)
        except BaseException as e:
            return _Seq(item.sequence_id, [], None, 1.0, error=e)
