"""Continuous batching for Qwen3-TTS: ``Qwen3TTSBatchSession`` with the protocol and the step semantics of the reference's
(``mlx_audio/tts/models/qwen3_tts/continuous_batching.py:37-360``), on slot KV caches instead of per-step cache merging.

``step()`` (continuous_batching.py:83-94): first every active request advances by ONE frame in one batched talker step, then -- if slots are free --
pending requests are admitted: prefilled as one left-padded batch and given their first frame in the same call.  A request leaves with a ``done``
event (its codes decoded by the speech tokenizer) when it samples EOS or reaches ``options.max_tokens`` frames; ``max_tokens <= 0`` answers every
request with an empty event without touching the model.  ``cancel`` drops a request wherever it is.

What differs from the reference is the storage, not the schedule: a request keeps ONE row ("slot") of every layer's KV buffer for its whole life
(``Qwen3TalkerSlots``, talker.py), so a step neither merges the active requests' caches into a padded batch nor extracts them again (two O(KV)
copies per frame in the reference); the device holds every request's history, trailing text and next input, and the only host
round trip of a step is the finished mask.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional

import torch

from ...continuous import TTSBatchEvent, TTSBatchItem, TTSBatchOptions


def _format_duration(seconds: float) -> str:
    hours = int(seconds // 3600)
    minutes = int((seconds % 3600) // 60)
    secs = seconds % 60
    return f"{hours:02d}:{minutes:02d}:{secs:06.3f}"


@dataclass
class _ActiveRequest:
    sequence_id: int
    text: str
    voice: Optional[str]
    instruct: Optional[str]
    slot: int
    frames: int = 0   # code frames generated so far (== len(generated_codes) of the reference's state)


class Qwen3TTSBatchSession:
    def __init__(self, model, options: TTSBatchOptions, *, slots=None, seed: Optional[int] = None):
        """``slots``: an object with the interface of ``Qwen3TalkerSlots`` (tests inject a scripted one); by default built on ``model.talker``."""
        if model.speech_tokenizer is None:
            raise ValueError("Speech tokenizer not loaded")
        self.model = model
        self.options = options
        self._pending: List[TTSBatchItem] = []
        self._active: List[_ActiveRequest] = []
        self._start_time = time.time()
        self._slots = slots
        self._seed = seed
        self.trace: Optional[list] = None   # tests: ("advance", [sequence ids]) / ("admit", [sequence ids]) in call order
        self.codes_log: Optional[dict] = None   # tests: sequence id -> the generated code frames [n, groups] behind its event

    # ------------------------------------------------------------------ protocol (tts/continuous.py:49-60)
    @property
    def idle(self) -> bool:
        return not self._pending and not self._active

    @property
    def available_slots(self) -> int:
        return max(0, self.options.max_batch_size - len(self._active))

    def add(self, items: List[TTSBatchItem]) -> None:
        self._pending.extend(items)

    def cancel(self, sequence_id: int) -> None:
        self._pending = [item for item in self._pending if item.sequence_id != sequence_id]
        for st in self._active:
            if st.sequence_id == sequence_id:
                self._engine().release(st.slot)
        self._active = [st for st in self._active if st.sequence_id != sequence_id]

    def step(self) -> List[TTSBatchEvent]:
        events: List[TTSBatchEvent] = []
        if self._active:
            events.extend(self._advance_active())
        if self.available_slots > 0 and self._pending:
            events.extend(self._admit_pending())
        return events

    # ------------------------------------------------------------------ engine
    def _engine(self):
        if self._slots is None:
            from .talker import Qwen3TalkerSlots

            o = self.options
            gen = None
            if o.temperature > 0:
                gen = torch.Generator(device=self.model.talker.device)
                gen.manual_seed(int(self._seed) if self._seed is not None else int(torch.seed() % (2 ** 31)))
            rows = self.model.talker.talker.cos.shape[0]
            self._slots = Qwen3TalkerSlots(self.model.talker, int(o.max_batch_size), max(1, min(int(o.max_tokens), rows - 1)), temperature=o.temperature,
                                           top_k=o.top_k, top_p=o.top_p, repetition_penalty=o.repetition_penalty, generator=gen)
        return self._slots

    def _free_slots(self) -> List[int]:
        used = {st.slot for st in self._active}
        return [s for s in range(int(self.options.max_batch_size)) if s not in used]

    # ------------------------------------------------------------------ admission (continuous_batching.py:109-189)
    def _admit_pending(self) -> List[TTSBatchEvent]:
        n = min(self.available_slots, len(self._pending))
        pending, self._pending = self._pending[:n], self._pending[n:]
        if not pending:
            return []
        if self.trace is not None:
            self.trace.append(("admit", [item.sequence_id for item in pending]))
        if self.options.max_tokens <= 0:
            return [self._empty_event(item.sequence_id) for item in pending]
        bi = self.model._prepare_batch_inputs([item.text for item in pending], language=self.options.lang_code, speakers=[item.voice for item in pending],
                                              instructs=[item.instruct for item in pending], return_metadata=True)
        eng = self._engine()
        slots = self._free_slots()[: len(pending)]
        finished = eng.admit(slots, bi.input_embeds, bi.left_padding, bi.trailing_text_hidden, bi.tts_pad_embed)
        events: List[TTSBatchEvent] = []
        for item, slot, fin in zip(pending, slots, finished):
            st = _ActiveRequest(item.sequence_id, item.text, item.voice, item.instruct, slot, frames=0 if fin else 1)
            if fin or st.frames >= min(self.options.max_tokens, eng.limit[slot]):
                events.append(self._decode_state(st))
            else:
                self._active.append(st)
        return events

    # ------------------------------------------------------------------ one frame for every active request (continuous_batching.py:191-239)
    def _advance_active(self) -> List[TTSBatchEvent]:
        eng = self._engine()
        states = list(self._active)
        if self.trace is not None:
            self.trace.append(("advance", [st.sequence_id for st in states]))
        finished = eng.advance(max(st.slot for st in states) + 1)
        events: List[TTSBatchEvent] = []
        still: List[_ActiveRequest] = []
        for st in states:
            fin = finished[st.slot]
            if not fin:
                st.frames += 1
            if fin or st.frames >= min(self.options.max_tokens, eng.limit[st.slot]):
                events.append(self._decode_state(st))
            else:
                still.append(st)
        self._active = still
        return events

    # ------------------------------------------------------------------ events (continuous_batching.py:322-360)
    def _decode_state(self, st: _ActiveRequest) -> TTSBatchEvent:
        eng = self._engine()
        if st.frames == 0:
            eng.release(st.slot)
            return self._empty_event(st.sequence_id)
        codes = eng.take_codes(st.slot, st.frames)
        eng.release(st.slot)
        if self.codes_log is not None:
            self.codes_log[st.sequence_id] = codes
        audio = self.model._decode_generated_codes(codes)
        dur = audio.shape[0] / self.model.sample_rate
        return TTSBatchEvent(sequence_id=st.sequence_id, audio=audio, sample_rate=self.model.sample_rate, samples=int(audio.shape[0]), token_count=st.frames,
                             done=True, metadata={"audio_duration": _format_duration(dur), "processing_time_seconds": time.time() - self._start_time,
                                                  "peak_memory_usage": torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0})

    def _empty_event(self, sequence_id: int) -> TTSBatchEvent:
        return TTSBatchEvent(sequence_id=sequence_id, audio=torch.zeros(0, dtype=torch.float32), sample_rate=self.model.sample_rate, samples=0, token_count=0,
                             done=True, metadata={"audio_duration": _format_duration(0.0), "processing_time_seconds": time.time() - self._start_time,
                                                  "peak_memory_usage": torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0})
