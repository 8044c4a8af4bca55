"""``POST /v1/audio/speech`` (mlx_audio/server.py:959-987) over this package's broker: the reference's request model, the broker's continuous-batching
session path and the serial ``generate`` path, audio bytes back.  Models are scripted stand-ins with the reference's model surface (no GPU)."""
import struct
import threading

import numpy as np
import pytest
import torch

fastapi = pytest.importorskip("fastapi")
from fastapi.testclient import TestClient  # noqa: E402


class SerialModel:
    """Only ``generate``: the broker's serial path (one request at a time)."""
    sample_rate = 24000

    def __init__(self):
        self.calls = []

    def generate(self, text, **kw):
        from types import SimpleNamespace

        self.calls.append((text, kw))
        n = 100 * len(text)
        yield SimpleNamespace(audio=torch.full((n,), 0.5), samples=n, sample_rate=24000)


class SessionModel(SerialModel):
    """Offers ``create_tts_batch_session``: requests in flight share the session's steps."""

    def supports_tts_continuous_batch(self, **kw):
        return kw.get("ref_audio") is None

    def create_tts_batch_session(self, options):
        from mlx_audio_amd.tts.continuous import TTSBatchEvent

        model = self
        model.options = options

        class Session:
            def __init__(self):
                self.items, self.steps = [], []

            @property
            def idle(self):
                return not self.items

            @property
            def available_slots(self):
                return options.max_batch_size - len(self.items)

            def add(self, items):
                self.items.extend(items)

            def cancel(self, sid):
                self.items = [i for i in self.items if i.sequence_id != sid]

            def step(self):
                self.steps.append([i.text for i in self.items])
                ev = [TTSBatchEvent(sequence_id=i.sequence_id, audio=torch.full((50 * len(i.text),), -0.25), sample_rate=24000, samples=50 * len(i.text), done=True)
                      for i in self.items]
                self.items = []
                return ev

        model.session = Session()
        return model.session


def _pcm(body):
    assert body[:4] == b"RIFF" and body[8:16] == b"WAVEfmt " and struct.unpack("<I", body[24:28])[0] == 24000
    return np.frombuffer(body[44:], dtype="<i2")


def test_speech_endpoint_serial_and_session_paths():
    from mlx_audio_amd.server import create_app

    serial, sess = SerialModel(), SessionModel()
    app = create_app({"serial": serial, "batched": sess}, max_batch_size=4)
    client = TestClient(app)
    try:
        assert [m["id"] for m in client.get("/v1/models").json()["data"]] == ["batched", "serial"]
        r = client.post("/v1/audio/speech", json={"model": "serial", "input": "hello", "voice": "af", "response_format": "wav", "speed": 1.2})
        assert r.status_code == 200 and r.headers["content-type"].startswith("audio/wav")
        pcm = _pcm(r.content)
        assert pcm.shape == (500,) and abs(int(pcm[0]) - 16383) <= 1
        text, kw = serial.calls[0]
        assert text == "hello" and kw["voice"] == "af" and kw["speed"] == 1.2 and kw["temperature"] == 0.7 and kw["lang_code"] == "a"   # SpeechRequest defaults
        # continuous-batching path: the session gets the request's sampling options (server.py:456-469) and returns raw pcm on request
        r = client.post("/v1/audio/speech", json={"model": "batched", "input": "four", "response_format": "pcm", "temperature": 0.3, "max_tokens": 77})
        assert r.status_code == 200 and np.frombuffer(r.content, dtype="<i2").shape == (200,)
        assert sess.options.temperature == 0.3 and sess.options.max_tokens == 77 and sess.options.max_batch_size == 4 and sess.session.steps == [["four"]]
        # concurrent requests are served (each gets its own audio back)
        out = {}

        def call(txt):
            out[txt] = client.post("/v1/audio/speech", json={"model": "batched", "input": txt, "response_format": "pcm"}).content

        ts = [threading.Thread(target=call, args=(t,)) for t in ("aa", "bbbb", "cccccc")]
        [t.start() for t in ts]
        [t.join(30) for t in ts]
        assert {k: len(v) // 2 for k, v in out.items()} == {"aa": 100, "bbbb": 200, "cccccc": 300}
        # errors of the shell itself
        assert client.post("/v1/audio/speech", json={"model": "nope", "input": "x"}).status_code == 404
        # the schema's default response_format is the reference's "mp3", for which no encoder is built: served as WAV, the substitution named in a header
        r = client.post("/v1/audio/speech", json={"model": "serial", "input": "x"})
        assert r.status_code == 200 and r.headers["content-type"].startswith("audio/wav") and r.headers["x-response-format-fallback"].startswith("mp3 -> wav")
        assert r.content[:4] == b"RIFF" and struct.unpack("<I", r.content[40:44])[0] == len(r.content) - 44    # non-streaming: a complete header
        assert client.post("/v1/audio/speech", json={"model": "serial", "input": "x", "response_format": "xyz"}).status_code == 400
        assert client.post("/v1/audio/speech", json={"model": "serial"}).status_code == 422                       # pydantic validation like the reference
    finally:
        app.state.broker.stop_and_join()


class FailingModel(SerialModel):
    def generate(self, text, **kw):
        from types import SimpleNamespace

        yield SimpleNamespace(audio=torch.full((100,), 0.25), samples=100, sample_rate=24000)
        raise RuntimeError("vocoder fell over")


def test_speech_endpoint_failures_are_5xx_when_nothing_was_sent():
    """A synthesis error of a non-streaming request must be an HTTP error, not a 200 with a truncated body (round-3 advisor finding)."""
    from mlx_audio_amd.server import create_app

    app = create_app({"bad": FailingModel(), "good": SerialModel()}, max_batch_size=2)
    client = TestClient(app, raise_server_exceptions=False)
    try:
        r = client.post("/v1/audio/speech", json={"model": "bad", "input": "hello", "response_format": "wav"})
        assert r.status_code == 500 and "vocoder fell over" in r.text
        r = client.post("/v1/audio/speech", json={"model": "good", "input": "hi", "response_format": "pcm"})   # the broker keeps serving
        assert r.status_code == 200 and len(r.content) == 2 * 200
    finally:
        app.state.broker.stop_and_join()
