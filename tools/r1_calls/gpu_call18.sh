#!/bin/bash
# round-1 call 18: ws3 producer straight-lining + conditional window loads + XCD-run tile order + interior epilogue / fold fast paths
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/t_conv18.log 2>&1
echo "conv+kokoro tests rc=$?" | tee -a $R
timeout 300 python tools/bench_conv.py --batch 32 --rounds 5 --out gpurun_out/conv_tile_ab_b32_v4.txt > gpurun_out/bench_conv18.log 2>&1
echo "bench_conv rc=$?" | tee -a $R
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench18_b64.json 2> gpurun_out/bench18_b64.err
echo "bench b64 rc=$?" | tee -a $R
timeout 300 python bench.py --batch 128 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench18_b128.json 2> gpurun_out/bench18_b128.err
echo "bench b128 rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_f" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_f.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_f.err"
echo "pmc fetch rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_w" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_w.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_w.err"
echo "pmc write rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DBF=$(find gpurun_out/pmc_f -name "*_results.db" | head -1); DBW=$(find gpurun_out/pmc_w -name "*_results.db" | head -1)
python tools/pmc_traffic.py "$DBF" "$DBW" > gpurun_out/hbm_traffic_kokoro_b64_v2.json 2> gpurun_out/pmc_traffic.err
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_k" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_k.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_k.err"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_k -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 4 | cut -c1-200 > gpurun_out/kokoro_kernel_stats_b64_v8.txt 2>&1; rm -rf gpurun_out/prof_k
cat $R; tail -n 8 gpurun_out/t_conv18.log; tail -n 30 gpurun_out/bench_conv18.log | cut -c1-150
python - <<'PY'
import json
for f in ("bench18_b64", "bench18_b128"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("conv_gemm_ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
d = json.load(open("gpurun_out/hbm_traffic_kokoro_b64_v2.json"))
for k, v in d["kernels"].items():
    if "conv_gemm" in k: print(k[:80], v["dispatches"], round(v["FETCH_SIZE_KB_avg"]), round(v["WRITE_SIZE_KB_avg"]))
PY
head -n 8 gpurun_out/kokoro_kernel_stats_b64_v8.txt
