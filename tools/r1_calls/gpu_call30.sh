#!/bin/bash
# round-1 call 30: the contract bench line again with the instrumented (roofline) step free of host synchronisation
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/bench_final30.json 2> gpurun_out/bench_final30.err; echo "bench rc=$?"
cat gpurun_out/bench_final30.json; tail -n 3 gpurun_out/bench_final30.err
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/bench_final30b.json 2> gpurun_out/bench_final30b.err; echo "bench(2) rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/bench_final30.json", "gpurun_out/bench_final30b.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"] / 1e6, 2), "M samples/s", round(d["ms_per_step"], 2), "ms; conv", round(r["conv_gemm_ms_per_step"], 2), "ms instrumented", round(r["instrumented_step_ms"], 2), "achieved", round(r["achieved"], 1), "frac", round(r["frac"], 4), "hbm frac", round(r["hbm_view"]["frac"], 4))
PY
