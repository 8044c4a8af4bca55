"""``bench.py --gpus N`` launches itself (torch.distributed.run, one rank per GPU): rehearsed here on CPU over gloo with a stand-in engine -- the
ranks, the rendezvous and every collective of the sharded step (mlx_audio_amd/shard.py) are the real ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_two_ranks_gloo():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-gloo", "--steps", "2", "--warmup", "1", "--batch", "3"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 6 and res["config"]["parallelism"] == "utterance-dp2"
    assert res["config"]["collectives_per_step"] >= 3          # broadcast, all_reduce of the frame counts, all_to_all of the waveforms
    assert res["value"] > 0 and "dry_run" in res
