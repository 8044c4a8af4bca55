#!/bin/bash
# round 5 call 25: the driver's round-end sequence on the build with ABI 33 (extrema partials; one-launch LSTM step; codec encode sides): full GPU suite, smoke(), the default bench command, the dsp line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_final.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $R
timeout 420 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?" >> $R
timeout 200 python bench.py --steps 20 --warmup 5 --gpus 1 > $O/bench_final_driver_flags.json 2> $O/bench_final2.err; echo "bench (driver flags) rc=$?" >> $R
timeout 120 python bench.py --config dsp > $O/bench_dsp_final.json 2>> $O/bench_final2.err; echo "bench dsp rc=$?" >> $R
cat $R; tail -4 $O/pytest_final.txt | cut -c1-200; tail -1 $O/smoke.txt; cut -c1-500 $O/bench_final.json; grep "bench +" $O/bench_final.err | cut -c1-120
python - <<'PY'
import json
for f in ("bench_final", "bench_final_driver_flags"):
    d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]
    print(f, round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 2), "ms p2", round(d.get("value_precision2", 0) / 1e6, 2), "frac", round(r["frac"], 4), "busy", r.get("mfma_busy_frac"), r.get("mfma_busy_frac_dominant_kernel"), "traffic x", r.get("traffic_over_algorithmic"), "lat", round(d["latency_b1"]["ms"], 2), "cpu", d["cpu_baseline"].get("value"), "check", d["batch_vs_single"]["max_abs_diff_over_peak"])
d = json.load(open("gpurun_out/bench_dsp_final.json")); print("dsp", round(d["ms_per_step"], 4), d["roofline"]["frac"], d.get("dense_input", {}).get("kernel_ms_per_step"))
PY
