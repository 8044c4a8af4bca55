#!/bin/bash
# round-2 call 1: ws4 (software-pipelined consumer, transposed epilogue, GEMM mode) against ws3 -- parity tests, per-shape A/B, the contract
# bench line, rocprofv3 kernel stats, and the FETCH_SIZE / WRITE_SIZE calibration on known byte counts
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --tb=short -p no:cacheprovider -k "conv" > gpurun_out/t_conv1.log 2>&1
echo "conv tests rc=$?" | tee -a $R
timeout 400 python tools/bench_conv.py --batch 32 --out gpurun_out/conv_ab_b32.txt > /dev/null 2> gpurun_out/conv_ab_b32.err
echo "bench_conv rc=$?" | tee -a $R
timeout 300 python tools/bench_conv.py --batch 64 --small --out gpurun_out/conv_small_b64.txt > /dev/null 2> gpurun_out/conv_small_b64.err
echo "bench_conv small rc=$?" | tee -a $R
timeout 300 python tools/bench_conv.py --batch 64 --small --flat --out gpurun_out/conv_small_flat_b64.txt > /dev/null 2> gpurun_out/conv_small_flat_b64.err
echo "bench_conv small flat rc=$?" | tee -a $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full1.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 600 python bench.py --shape-table gpurun_out/shape_table_ws4.txt > gpurun_out/bench1.json 2> gpurun_out/bench1.err
echo "bench rc=$?" | tee -a $R
MI355_CONV_WS_VARIANT=7 timeout 600 python bench.py --no-cpu-baseline --shape-table gpurun_out/shape_table_ws3.txt > gpurun_out/bench1_ws3.json 2> gpurun_out/bench1_ws3.err
echo "bench ws3 rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_k" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_k.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_k.err"
echo "rocprof kokoro rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/cal_f" -o f -- "$GRAFT_REPO_ROOT/tools/bin/pmc_calib" > "$GRAFT_REPO_ROOT/gpurun_out/calib_f.log" 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/cal_w" -o w -- "$GRAFT_REPO_ROOT/tools/bin/pmc_calib" > "$GRAFT_REPO_ROOT/gpurun_out/calib_w.log" 2>&1
echo "calib rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_k -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 4 | cut -c1-200 > gpurun_out/kokoro_kernel_stats_c1.txt 2>&1; rm -rf gpurun_out/prof_k
FDB=$(find gpurun_out/cal_f -name "*_results.db" | head -1); WDB=$(find gpurun_out/cal_w -name "*_results.db" | head -1)
python tools/pmc_traffic.py "$FDB" "$WDB" copy > gpurun_out/pmc_calibration.json 2> gpurun_out/pmc_calibration.err; rm -rf gpurun_out/cal_f gpurun_out/cal_w
cat $R; tail -n 15 gpurun_out/t_conv1.log | cut -c1-250; tail -n 12 gpurun_out/t_full1.log | cut -c1-250
cat gpurun_out/conv_ab_b32.txt; cat gpurun_out/conv_small_b64.txt; cat gpurun_out/conv_small_flat_b64.txt
cat gpurun_out/bench1.json | cut -c1-1800; tail -n 3 gpurun_out/bench1.err; cat gpurun_out/bench1_ws3.json | cut -c1-600
head -n 16 gpurun_out/kokoro_kernel_stats_c1.txt | cut -c1-160
grep "rep 2" gpurun_out/calib_f.log; python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_calibration.json"))["kernels"]
for k,v in d.items(): print(k[:40], "FETCH KB", round(v["FETCH_SIZE_KB_avg"]), "WRITE KB", round(v["WRITE_SIZE_KB_avg"]), "(true: 1048576 KB each)")
PY
