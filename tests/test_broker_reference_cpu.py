"""The serving shell's scheduler against the reference's own: ``tests/golden/ref_broker.json`` is what the reference's ``InferenceBroker``
(server_inference.py:129-358, imported by tests/golden/make_reference_fixtures.py) does with the scripted scenario of ``pt_layouts.broker_scenario`` -- the
worker is held inside the first request while sixteen more arrive, then sees them all at once.  This package's broker, given the same scenario, must
make the same adapter / session calls in the same order (continuous sessions first and until they drain, then fixed-window batches of at most
``max_batch_size`` compatible requests, serial calls for the rest) and deliver the same result chunks to every request."""
import json
import os
import sys

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import pt_layouts as PT  # noqa: E402


def test_broker_schedules_the_scenario_like_the_reference():
    from mlx_audio_amd import server_inference as mod

    want = json.load(open(os.path.join(GOLD, "ref_broker.json")))
    got = json.loads(json.dumps(PT.broker_scenario(mod)))
    assert got["unknown_endpoint"] == want["unknown_endpoint"]
    assert got["trace"] == want["trace"], (got["trace"], want["trace"])
    assert got["chunks"] == want["chunks"]
    # what the scenario is there to show (read off the reference's trace)
    t = want["trace"]
    assert ["batch", [1, 3, 5]] in t and ["batch", [6, 12]] in t and ["serial", 4] in t and ["serial", 16] in t
    assert t.index(["step", "X", [9]]) < t.index(["batch", [1, 3, 5]])                     # sessions drain before any window runs
    assert want["chunks"]["7"] == [] and want["chunks"]["11"][0][0] == "error" and want["chunks"]["14"][0] == ["error", "session refuses this request"]
