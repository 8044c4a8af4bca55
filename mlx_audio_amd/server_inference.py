"""Inference broker: the one worker thread through which a serving shell funnels every model call (``mlx_audio/server_inference.py:14-358``).

Same interface as the reference (``InferenceRequest`` / ``InferenceHandle`` / ``InferenceResultChunk`` / ``InferenceContext``, the
``ModelExecutionAdapter`` and ``ContinuousBatchSession`` protocols, ``InferenceBroker.register_adapter / submit / stop_and_join``), so an HTTP front end
written against it keeps working.  What differs is underneath: the reference gives its worker thread-local MLX streams (``:195-202``); here the worker
owns one HIP stream (``torch.cuda.Stream``) and every kernel of every request is enqueued on it, and the TTS adapter below drives the batched MI355X
engines through the ``TTSBatchSession`` protocol (``tts/continuous.py``) instead of looping over requests one by one.

Scheduling rules kept from the reference (``:205-262``): all GPU work is serialised on the worker; requests whose adapter offers a continuous session
are routed to the session of their (endpoint, model, batch key) and advance one ``step()`` per loop turn, joining between steps; whole-request serial
or fixed-window batch calls only run while no continuous session is active; a cancelled request is dropped before it starts; an exception inside a
call is delivered to the affected requests as an ``error`` chunk followed by ``done`` and never kills the worker.
"""
from __future__ import annotations

import queue
import threading
import time
import traceback
import uuid
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Protocol

import torch


# ------------------------------------------------------------------------------------------------ wire types (server_inference.py:14-64)
@dataclass
class InferenceResultChunk:
    kind: str                      # "data" | "error" | "done"
    payload: Any = None
    error: Optional[BaseException] = None


@dataclass
class InferenceContext:
    request_id: str
    endpoint_kind: str
    model_name: str
    queued_at: float
    batch_key: Any = None


@dataclass
class InferenceRequest:
    endpoint_kind: str
    model_name: str
    payload: Any
    normalized_kwargs: Dict[str, Any] = field(default_factory=dict)
    stream: bool = False
    batch_key: Any = None
    request_id: str = field(default_factory=lambda: uuid.uuid4().hex)
    queued_at: float = field(default_factory=time.time)
    result_queue: "queue.Queue[InferenceResultChunk]" = field(default_factory=queue.Queue)
    cancel_event: threading.Event = field(default_factory=threading.Event)

    def emit_data(self, payload: Any) -> None:
        self.result_queue.put(InferenceResultChunk("data", payload=payload))

    def emit_error(self, error: BaseException) -> None:
        self.result_queue.put(InferenceResultChunk("error", error=error))

    def emit_done(self) -> None:
        self.result_queue.put(InferenceResultChunk("done"))


@dataclass
class InferenceHandle:
    context: InferenceContext
    result_queue: "queue.Queue[InferenceResultChunk]"
    cancel_event: threading.Event

    def cancel(self) -> None:
        self.cancel_event.set()


class ContinuousBatchSession(Protocol):
    @property
    def idle(self) -> bool: ...

    def submit(self, request: InferenceRequest) -> None: ...

    def step(self) -> None: ...

    def fail(self, error: BaseException) -> None: ...


class ModelExecutionAdapter(Protocol):
    max_batch_size: int

    def supports_batch(self, request: InferenceRequest) -> bool: ...

    def batch_key(self, request: InferenceRequest) -> Any: ...

    def run_serial(self, request: InferenceRequest) -> None: ...

    def run_batch(self, requests: List[InferenceRequest]) -> None: ...

    def supports_continuous_batch(self, request: InferenceRequest) -> bool: ...

    def continuous_batch_key(self, request: InferenceRequest) -> Any: ...

    def create_continuous_batch_session(self, request: InferenceRequest) -> ContinuousBatchSession: ...


class BaseModelExecutionAdapter:
    """Defaults of the adapter protocol (``:100-126``): serial only."""

    max_batch_size = 1

    def supports_batch(self, request: InferenceRequest) -> bool:
        return False

    def batch_key(self, request: InferenceRequest) -> Any:
        return None

    def run_serial(self, request: InferenceRequest) -> None:
        raise NotImplementedError

    def run_batch(self, requests: List[InferenceRequest]) -> None:
        if len(requests) != 1:
            raise NotImplementedError
        self.run_serial(requests[0])

    def supports_continuous_batch(self, request: InferenceRequest) -> bool:
        return False

    def continuous_batch_key(self, request: InferenceRequest) -> Any:
        return self.batch_key(request)

    def create_continuous_batch_session(self, request: InferenceRequest) -> ContinuousBatchSession:
        raise NotImplementedError


# ------------------------------------------------------------------------------------------------ the broker
class InferenceBroker:
    def __init__(self, *, idle_poll_s: float = 0.1, device: Optional[str] = None):
        self.idle_poll_s = idle_poll_s
        self.device = device
        self._inbox: "queue.Queue[Optional[InferenceRequest]]" = queue.Queue()
        self._adapters: Dict[str, ModelExecutionAdapter] = {}
        self._sessions: Dict[Any, ContinuousBatchSession] = {}
        self._halt = threading.Event()
        self._worker = threading.Thread(target=self._worker_main, name="mi355-inference-broker", daemon=True)
        self._worker.start()

    # -- public surface (server_inference.py:144-193)
    def register_adapter(self, endpoint_kind: str, adapter: ModelExecutionAdapter) -> None:
        self._adapters[endpoint_kind] = adapter

    def submit(self, *, endpoint_kind: str, model_name: str, payload: Any, normalized_kwargs: Optional[Dict[str, Any]] = None, stream: bool = False,
               batch_key: Any = None) -> InferenceHandle:
        adapter = self._adapters.get(endpoint_kind)
        if adapter is None:
            raise ValueError(f"No inference adapter registered for {endpoint_kind!r}")
        req = InferenceRequest(endpoint_kind=endpoint_kind, model_name=model_name, payload=payload, normalized_kwargs=normalized_kwargs or {},
                               stream=stream, batch_key=batch_key)
        if req.batch_key is None:
            req.batch_key = adapter.batch_key(req)
        self._inbox.put(req)
        ctx = InferenceContext(req.request_id, req.endpoint_kind, req.model_name, req.queued_at, req.batch_key)
        return InferenceHandle(ctx, req.result_queue, req.cancel_event)

    def stop_and_join(self, timeout: float = 5.0) -> None:
        self._halt.set()
        self._inbox.put(None)
        self._worker.join(timeout=timeout)
        for adapter in self._adapters.values():
            shutdown = getattr(adapter, "shutdown", None)
            if callable(shutdown):
                shutdown()

    # -- worker
    def _worker_main(self) -> None:
        # one HIP stream for everything this worker launches (the reference creates thread-local MLX streams here, :195-202)
        if torch.cuda.is_available():
            stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(stream):
                self._loop()
        else:
            self._loop()

    def _loop(self) -> None:
        waiting: List[InferenceRequest] = []
        try:
            while not self._halt.is_set():
                self._intake(waiting, block=not waiting and not self._sessions)
                waiting[:] = [r for r in waiting if not r.cancel_event.is_set()]
                self._feed_sessions(waiting)
                self._advance_sessions()
                if self._sessions or not waiting:
                    continue            # continuous sessions own the GPU until they drain
                self._run_window(waiting)
        finally:
            for s in list(self._sessions.values()):
                s.fail(RuntimeError("Inference broker stopped."))
            self._sessions.clear()

    def _intake(self, waiting: List[InferenceRequest], *, block: bool) -> None:
        first = True
        while True:
            try:
                item = self._inbox.get(timeout=self.idle_poll_s) if (block and first) else self._inbox.get_nowait()
            except queue.Empty:
                return
            first = False
            if item is None:
                self._halt.set()
                return
            waiting.append(item)

    def _feed_sessions(self, waiting: List[InferenceRequest]) -> None:
        keep: List[InferenceRequest] = []
        for req in waiting:
            adapter = self._adapters.get(req.endpoint_kind)
            if adapter is None or not adapter.supports_continuous_batch(req):
                keep.append(req)
                continue
            key = (req.endpoint_kind, req.model_name, adapter.continuous_batch_key(req))
            try:
                session = self._sessions.get(key)
                if session is None or session.idle:
                    session = adapter.create_continuous_batch_session(req)
                    self._sessions[key] = session
                session.submit(req)
            except Exception as exc:
                traceback.print_exc()
                req.emit_error(exc)
                req.emit_done()
        waiting[:] = keep

    def _advance_sessions(self) -> None:
        for key, session in list(self._sessions.items()):
            try:
                session.step()
            except Exception as exc:
                traceback.print_exc()
                session.fail(exc)
                self._sessions.pop(key, None)
                continue
            if session.idle:
                self._sessions.pop(key, None)

    def _run_window(self, waiting: List[InferenceRequest]) -> None:
        head = waiting.pop(0)
        adapter = self._adapters.get(head.endpoint_kind)
        if adapter is None:
            head.emit_error(ValueError(f"No inference adapter registered for {head.endpoint_kind!r}"))
            head.emit_done()
            return
        group = [head]
        if adapter.supports_batch(head) and adapter.max_batch_size > 1:
            rest: List[InferenceRequest] = []
            for cand in waiting:
                same = (cand.endpoint_kind == head.endpoint_kind and cand.model_name == head.model_name and cand.batch_key == head.batch_key
                        and not cand.cancel_event.is_set())
                if same and len(group) < adapter.max_batch_size and adapter.supports_batch(cand):
                    group.append(cand)
                else:
                    rest.append(cand)
            waiting[:] = rest
        try:
            if len(group) > 1:
                adapter.run_batch(group)
            else:
                adapter.run_serial(head)
        except Exception as exc:
            traceback.print_exc()
            for r in group:
                r.emit_error(exc)
                r.emit_done()


# ------------------------------------------------------------------------------------------------ TTS adapter over the batch-session protocol
class _SpeechSession:
    """``ContinuousBatchSession`` over a model's ``TTSBatchSession`` (the role of the reference's ``_TTSAdapterContinuousSession``, server.py:585-600):
    requests beyond the session's free slots wait in a backlog and join as slots free up; every ``TTSBatchEvent`` becomes a ``data`` chunk on its
    request (``error`` for failed items), finished requests get ``done``; a cancelled request leaves the model session at the next step."""

    def __init__(self, model_session):
        self.inner = model_session
        self._by_id: Dict[int, InferenceRequest] = {}
        self._backlog: List[InferenceRequest] = []
        self._next_id = 0

    @property
    def idle(self) -> bool:
        return self.inner.idle and not self._backlog and not self._by_id

    def submit(self, request: InferenceRequest) -> None:
        self._backlog.append(request)

    def _admit(self) -> None:
        from .tts.continuous import TTSBatchItem

        while self._backlog and self.inner.available_slots > 0:
            req = self._backlog.pop(0)
            if req.cancel_event.is_set():
                req.emit_done()
                continue
            p = req.payload if isinstance(req.payload, dict) else {"text": str(req.payload)}
            sid, self._next_id = self._next_id, self._next_id + 1
            self._by_id[sid] = req
            self.inner.add([TTSBatchItem(sequence_id=sid, text=p.get("text", ""), voice=p.get("voice"), speed=p.get("speed"), instruct=p.get("instruct"),
                                         extra={k: p[k] for k in ("lang_code", "split_pattern") if k in p})])

    def step(self) -> None:
        for sid, req in list(self._by_id.items()):
            if req.cancel_event.is_set():
                self.inner.cancel(sid)
                self._by_id.pop(sid)
                req.emit_done()
        self._admit()
        if self.inner.idle:
            return
        for ev in self.inner.step():
            req = self._by_id.get(ev.sequence_id)
            if req is None:
                continue
            if ev.error is not None:
                req.emit_error(ev.error)
            else:
                req.emit_data(ev)
            if ev.done:
                self._by_id.pop(ev.sequence_id, None)
                req.emit_done()

    def fail(self, error: BaseException) -> None:
        for req in list(self._by_id.values()) + self._backlog:
            req.emit_error(error)
            req.emit_done()
        self._by_id.clear()
        self._backlog.clear()


class TTSExecutionAdapter(BaseModelExecutionAdapter):
    """Endpoint adapter for text-to-speech models of this build.  ``payload`` is a dict: ``text`` (required), ``voice``, ``speed``, ``lang_code``,
    ``instruct``, ``ref_audio``, ``ref_text``, ``pitch``.  Models that offer ``create_tts_batch_session`` (Kokoro) are served by continuous batching;
    others fall back to ``model.generate`` per request.  ``models`` maps the request's ``model_name`` to a loaded model (the reference resolves it
    through its model provider, server.py:470-483)."""

    def __init__(self, models: Dict[str, Any], max_batch_size: int = 8):
        self.models = models
        self.max_batch_size = max_batch_size

    def _model(self, request: InferenceRequest):
        try:
            return self.models[request.model_name]
        except KeyError:
            raise ValueError(f"unknown model {request.model_name!r}") from None

    @staticmethod
    def _probe_kwargs(request: InferenceRequest) -> Dict[str, Any]:
        p = request.payload if isinstance(request.payload, dict) else {}
        return dict(stream=request.stream, voice=p.get("voice"), instruct=p.get("instruct"), ref_audio=p.get("ref_audio"), ref_text=p.get("ref_text"),
                    speed=p.get("speed", 1.0), pitch=p.get("pitch", 1.0))

    def batch_key(self, request: InferenceRequest) -> Any:
        p = request.payload if isinstance(request.payload, dict) else {}
        return ("tts", request.stream, p.get("lang_code", "a"))

    def supports_continuous_batch(self, request: InferenceRequest) -> bool:
        model = self.models.get(request.model_name)
        hook = getattr(model, "supports_tts_continuous_batch", None)
        return callable(hook) and callable(getattr(model, "create_tts_batch_session", None)) and bool(hook(**self._probe_kwargs(request)))

    def create_continuous_batch_session(self, request: InferenceRequest) -> ContinuousBatchSession:
        from .tts.continuous import TTSBatchOptions

        p = request.payload if isinstance(request.payload, dict) else {}
        # server.py:456-469: the session is created with the FIRST request's sampling options (requests of one session share a batch key)
        defaults = TTSBatchOptions()
        opts = TTSBatchOptions(temperature=p.get("temperature", defaults.temperature), top_p=p.get("top_p", defaults.top_p), top_k=p.get("top_k", defaults.top_k),
                               repetition_penalty=p.get("repetition_penalty", defaults.repetition_penalty), max_tokens=p.get("max_tokens", defaults.max_tokens),
                               lang_code=p.get("lang_code", "a"), stream=request.stream, streaming_interval=p.get("streaming_interval", defaults.streaming_interval),
                               max_batch_size=self.max_batch_size, verbose=bool(p.get("verbose", False)))
        return _SpeechSession(self._model(request).create_tts_batch_session(opts))

    def run_serial(self, request: InferenceRequest) -> None:
        p = request.payload if isinstance(request.payload, dict) else {"text": str(request.payload)}
        kw = {k: v for k, v in p.items() if k != "text" and v is not None}
        try:
            for result in self._model(request).generate(p.get("text", ""), **kw):
                if request.cancel_event.is_set():
                    break
                request.emit_data(result)
        except Exception as exc:
            request.emit_error(exc)
        request.emit_done()
