"""the serving shell wired to the multi-GPU engine.  The HTTP endpoint, the ``InferenceBroker`` and the ``KokoroBatchSession`` run on
rank 0 and call ``shard.ShardedKokoro.forward`` exactly as they call the single-GPU engine; the other ranks sit in ``worker_loop()``.  Exercised over
``gloo`` at world 2 and 3 with the stand-in engine of tests/test_shard_cpu.py (same ``front`` / ``back`` contract as the Kokoro engine) and a scripted
model surface (pipeline, voice pack, vocabulary) -- no GPU, no G2P."""
import os
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_amd import shard
from mlx_audio_amd.tts.continuous import TTSBatchItem, TTSBatchOptions
from mlx_audio_amd.tts.models.kokoro.continuous_batching import KokoroBatchSession
from test_shard_cpu import SPF, FakeEngine, _free_port


class _SingleEngine(FakeEngine):
    """The stand-in as ONE process runs it: ``forward`` = both halves back to back (what the real engine's ``forward`` is)."""

    def forward(self, ids, ref_s, speed=1.0):
        return self.back(self.front(ids, ref_s, speed=speed))


class _Pipeline:
    """Scripted G2P: a 'phoneme' per letter, chunks of <= 12 symbols; voice pack rows depend on the chunk length like the reference's."""

    def load_voice(self, voice):
        base = float(sum(map(ord, voice)) % 17)
        return torch.arange(510, dtype=torch.float32)[:, None, None] * 0.01 + base + torch.zeros(510, 1, 2 * FakeEngine.sty)

    def phoneme_chunks(self, text):
        ps = "".join(c for c in text.lower() if c.isalpha())
        for i in range(0, len(ps), 12):
            yield text, ps[i:i + 12], None


class _Model:
    sample_rate = 24000
    vocab = {c: i + 1 for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}

    def __init__(self, engine):
        self.engine = engine
        self._p = _Pipeline()

    def _get_pipeline(self, lang):
        return self._p

    def phonemes_to_ids(self, ps):
        return torch.tensor([0] + [self.vocab[p] for p in ps if p in self.vocab] + [0])

    def supports_tts_continuous_batch(self, **kw):
        return True

    def create_tts_batch_session(self, options):
        return KokoroBatchSession(self, options)


TEXTS = ["hello there general", "a", "the quick brown fox jumps over the lazy dog", "speech", "one two three four five six seven"]


def _drive(model, speeds):
    """All requests through one session; returns {sequence_id: waveform}."""
    sess = model.create_tts_batch_session(TTSBatchOptions(max_batch_size=8))
    sess.add([TTSBatchItem(sequence_id=i, text=t, voice="af_heart" if i % 2 else "bm_x", speed=speeds[i]) for i, t in enumerate(TEXTS)])
    out = {}
    for _ in range(64):
        for ev in sess.step():
            assert ev.error is None, ev.error
            if ev.done:
                out[ev.sequence_id] = ev.audio
        if sess.idle:
            break
    return out


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ch = shard.ShardChannel("cpu", dist, max_items=16, max_tokens=32)
        sk = shard.ShardedKokoro(FakeEngine(skew=True), ch, SPF)
        if rank != 0:
            served = sk.worker_loop()
            q.put(("worker", rank, served, sk.engine.front_calls))
            return
        model = _Model(sk)
        speeds = [1.0, 1.0, 1.25, 1.0, 1.25]
        if mode == "session":
            got = _drive(model, speeds)
            want = _drive(_Model(_SingleEngine(skew=True)), speeds)
            ok = sorted(got) == sorted(want) == list(range(len(TEXTS))) and all(torch.equal(got[i], want[i]) for i in want)
        else:  # the HTTP endpoint -> broker -> session -> sharded engine, three requests in flight
            from fastapi.testclient import TestClient

            from mlx_audio_amd.server import create_app

            app = create_app({"kokoro": model}, max_batch_size=4)
            client = TestClient(app)
            res = {}

            def call(i):
                res[i] = client.post("/v1/audio/speech", json={"model": "kokoro", "input": TEXTS[i], "voice": "af_heart", "response_format": "pcm"})

            try:
                ts = [threading.Thread(target=call, args=(i,)) for i in (0, 2, 3)]
                [t.start() for t in ts]
                [t.join(60) for t in ts]
            finally:
                app.state.broker.stop_and_join()
            single = _Model(_SingleEngine(skew=True))
            ok = True
            for i in (0, 2, 3):
                sess = single.create_tts_batch_session(TTSBatchOptions(max_batch_size=4))
                sess.add([TTSBatchItem(sequence_id=0, text=TEXTS[i], voice="af_heart", speed=1.0)])
                audio = None
                while not sess.idle:
                    for ev in sess.step():
                        audio = ev.audio if ev.done else audio
                pcm = np.frombuffer(res[i].content, dtype="<i2")
                want = np.clip(np.round(audio.numpy().clip(-1, 1) * 32767.0), -32768, 32767).astype(np.int16)
                ok = ok and res[i].status_code == 200 and pcm.shape == want.shape and int(np.abs(pcm.astype(np.int32) - want).max()) <= 1
        sk.close()
        q.put(("src", ok, sk.steps, ch.collectives))
    finally:
        dist.destroy_process_group()


def _run(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return [q.get() for _ in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_batch_session_over_sharded_engine(world):
    res = _run(world, "session")
    src = [r for r in res if r[0] == "src"][0]
    workers = [r for r in res if r[0] == "worker"]
    assert src[1], "waveforms of the sharded session differ from the single-process session"
    assert src[2] >= 4 and len(workers) == world - 1      # 4 chunks of the longest text = at least 4 steps (two speed groups: more)
    assert all(w[2] == src[2] for w in workers)           # every worker served every step
    assert sum(w[3] for w in workers) > 0                 # ... and ran the token-rate half of its share


def test_speech_endpoint_over_sharded_engine():
    pytest.importorskip("fastapi")
    res = _run(2, "http")
    src = [r for r in res if r[0] == "src"][0]
    assert src[1] and src[2] >= 1
