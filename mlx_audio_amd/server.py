"""The OpenAI-style speech endpoint of the reference's serving shell (``mlx_audio/server.py:959-987``: ``POST /v1/audio/speech``) over this package's
``InferenceBroker`` -- the thinnest shell that lets a client of the reference switch: the request model (``SpeechRequest``, server.py:167-186: same fields and
defaults), submission to the broker (endpoint kind "tts": continuous-batching sessions for Kokoro / Qwen3-TTS, serial ``generate`` otherwise), and a streaming
response that carries each result chunk as audio bytes.  ``GET /v1/models`` lists what is loaded.

Deliberately NOT here (control plane of the reference, out of scope of the hot path: SURVEY section 8 / DESIGN section 8): model download / hot loading, the
web UI, CORS / auth, STT / separation endpoints, mp3 / flac / opus containers (``response_format`` "wav" and "pcm" only: the reference encodes the others with
ffmpeg-backed writers).  Models are handed in already loaded (``create_app({"name": model})``).

Multi-GPU (one process per GPU, ``torch.distributed`` over RCCL): ``attach_sharded_engine(model, dist)`` on every rank.  Rank 0 then serves
``create_app({...: model})`` as on one GPU -- the broker's ``KokoroBatchSession`` calls ``model.engine.forward`` for whatever requests are in flight, and that
call is now ``shard.ShardedKokoro.forward``: request block and style rows out, token-rate half on the token-LPT shard, re-balance on the real frame counts,
frame-rate half, waveforms back -- while the other ranks block in the returned object's ``worker_loop()`` until rank 0 calls ``close()``
(tests/test_shard_serving_cpu.py: the endpoint over gloo at world 2).  The autoregressive families shard by sequence with ``shard.sharded_decode``.
"""
from __future__ import annotations

import io
from typing import Any, Dict, Optional

from .server_inference import InferenceBroker, TTSExecutionAdapter


from pydantic import BaseModel  # noqa: E402  (fastapi's dependency; the endpoint parameter's annotation must resolve at module level)


class SpeechRequest(BaseModel):   # server.py:167-186
    model: str
    input: str
    instruct: Optional[str] = None
    voice: Optional[str] = None
    speed: Optional[float] = 1.0
    gender: Optional[str] = "male"
    pitch: Optional[float] = 1.0
    lang_code: Optional[str] = "a"
    ref_audio: Optional[str] = None
    ref_text: Optional[str] = None
    temperature: Optional[float] = 0.7
    top_p: Optional[float] = 0.95
    top_k: Optional[int] = 40
    repetition_penalty: Optional[float] = 1.0
    response_format: Optional[str] = "mp3"
    stream: bool = False
    streaming_interval: float = 2.0
    max_tokens: int = 1200
    verbose: bool = False



def _pcm16(audio) -> bytes:
    import numpy as np
    import torch

    a = audio.detach().to(torch.float32).cpu().numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio, dtype=np.float32)
    return (np.clip(a.reshape(-1), -1.0, 1.0) * 32767.0).astype("<i2").tobytes()


def _wav_header(sample_rate: int, n_bytes: int = 0xFFFFFFFF - 36) -> bytes:
    """RIFF header of a 16-bit mono stream; with the size unknown up front (streaming) the placeholder length players accept."""
    import struct

    return b"RIFF" + struct.pack("<I", (36 + n_bytes) & 0xFFFFFFFF) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + \
        b"data" + struct.pack("<I", n_bytes & 0xFFFFFFFF)


def create_app(models: Dict[str, Any], *, max_batch_size: int = 8, broker: Optional[InferenceBroker] = None):
    """FastAPI app serving ``models`` (name -> loaded model with the reference's ``generate`` / ``create_tts_batch_session`` surface)."""
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import StreamingResponse

    app = FastAPI(title="mlx_audio_amd")
    brk = broker or InferenceBroker()
    brk.register_adapter("tts", TTSExecutionAdapter(models, max_batch_size=max_batch_size))
    app.state.broker = brk

    @app.get("/v1/models")
    def list_models():
        return {"object": "list", "data": [{"id": name, "object": "model"} for name in sorted(models)]}

    @app.post("/v1/audio/speech")
    def tts_speech(payload: SpeechRequest):
        if payload.model not in models:
            raise HTTPException(status_code=404, detail=f"Model '{payload.model}' is not loaded")
        asked = (payload.response_format or "wav").lower()
        # the schema keeps the reference's default "mp3" (server.py:959 of the reference), but no compressed encoder is built here: a request
        # for mp3 / opus / aac / flac -- including every OpenAI-style call that simply omits response_format -- is answered as WAV, with the substitution
        # named in a response header, instead of failing
        compressed = ("mp3", "opus", "aac", "flac")
        if asked not in ("wav", "pcm") + compressed:
            raise HTTPException(status_code=400, detail=f"unknown response_format '{asked}' (wav | pcm; {' | '.join(compressed)} are served as wav)")
        fmt = "wav" if asked in compressed else asked
        body = payload.model_dump(exclude={"model", "input", "response_format"}, exclude_none=True)
        body["text"] = payload.input
        handle = brk.submit(endpoint_kind="tts", model_name=payload.model, payload=body, normalized_kwargs=body, stream=payload.stream)
        sr = int(getattr(models[payload.model], "sample_rate", 24000))

        headers = {"Content-Disposition": f"attachment; filename=speech.{fmt}"}
        if fmt != asked:
            headers["X-Response-Format-Fallback"] = f"{asked} -> wav (no {asked} encoder in this build)"

        def cancel():
            c = getattr(handle, "cancel", None)
            if callable(c):
                c()

        if not payload.stream:
            # nothing has been sent yet: collect the audio first, so that a failure is an HTTP 5xx and not a 200 with a truncated body
            parts = []
            try:
                while True:
                    chunk = handle.result_queue.get()
                    if chunk.kind == "done":
                        break
                    if chunk.kind == "error":
                        raise HTTPException(status_code=500, detail=f"synthesis failed: {chunk.error}")
                    audio = getattr(chunk.payload, "audio", chunk.payload)
                    if audio is not None:
                        parts.append(_pcm16(audio))
            except BaseException:
                cancel()
                raise
            body_bytes = b"".join(parts)
            if fmt == "wav":
                body_bytes = _wav_header(sr, len(body_bytes)) + body_bytes
            from fastapi.responses import Response

            return Response(content=body_bytes, media_type=f"audio/{fmt}", headers=headers)

        def chunks():
            first, finished = True, False
            try:
                while True:
                    chunk = handle.result_queue.get()
                    if chunk.kind == "done":
                        finished = True
                        break
                    if chunk.kind == "error":
                        finished = True
                        raise RuntimeError(str(chunk.error))   # bytes may already be out: the connection is dropped mid-body, the slot is released below
                    audio = getattr(chunk.payload, "audio", chunk.payload)
                    if audio is None:
                        continue
                    data = _pcm16(audio)
                    if first and fmt == "wav":
                        yield _wav_header(sr)
                    first = False
                    yield data
                if first and fmt == "wav":
                    yield _wav_header(sr, 0)
            finally:
                if not finished:   # client went away (generator closed) or the stream broke: do not keep the batch slot busy
                    cancel()

        return StreamingResponse(chunks(), media_type=f"audio/{fmt}", headers=headers)

    return app


def attach_sharded_engine(model, dist, *, max_items: int = 1024, max_tokens: int = 512, wire_dtype=None):
    """Every rank calls this with its own loaded copy of a Kokoro-family ``model`` (engine with ``front`` / ``back``).  Replaces ``model.engine`` by a
    ``shard.ShardedKokoro`` over all ranks and returns it: rank 0 goes on to ``create_app`` / ``uvicorn.run`` and calls ``.close()`` at shutdown, the other
    ranks call ``.worker_loop()``."""
    from . import shard

    eng = model.engine
    ch = shard.ShardChannel(eng.dev if hasattr(eng, "dev") else "cpu", dist, max_items=max_items, max_tokens=max_tokens)
    spf = 2 * int(eng.total_up)   # samples per duration frame: 2 x prod(upsample rates) x iSTFT hop (Kokoro: 600)
    sk = shard.ShardedKokoro(eng, ch, spf, wire_dtype=wire_dtype)
    model.engine = sk
    return sk
