#!/bin/bash
# round-1 call 11: the complete GPU suite in one process (what the driver runs), smoke(), the contract bench line with cpu_baseline,
# rocprofv3 kernel stats of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) for roofline.traffic
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full11.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke11.log 2>&1
echo "smoke rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_f" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_f.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_f.err"
echo "pmc fetch rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_w" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_w.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_w.err"
echo "pmc write rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DBF=$(find gpurun_out/pmc_f -name "*_results.db" | head -1); DBW=$(find gpurun_out/pmc_w -name "*_results.db" | head -1)
python tools/pmc_traffic.py "$DBF" "$DBW" > gpurun_out/hbm_traffic_kokoro_b64.json 2> gpurun_out/pmc_traffic.err
echo "pmc summary rc=$?" | tee -a $R
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
mkdir -p profiles; cp gpurun_out/hbm_traffic_kokoro_b64.json profiles/r1_hbm_traffic_kokoro_b64.json
timeout 600 python bench.py > gpurun_out/bench_final11.json 2> gpurun_out/bench_final11.err
echo "bench rc=$?" | tee -a $R
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_k" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_k.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_k.err"
echo "rocprof kokoro rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_k -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 4 | cut -c1-200 > gpurun_out/kokoro_kernel_stats_v5.txt 2>&1; rm -rf gpurun_out/prof_k
cat $R; tail -n 15 gpurun_out/t_full11.log; tail -n 3 gpurun_out/smoke11.log; cat gpurun_out/bench_final11.json; tail -n 3 gpurun_out/bench_final11.err; head -c 300 gpurun_out/hbm_traffic_kokoro_b64.json; tail -n 5 gpurun_out/pmc_traffic.err gpurun_out/pmc_f.err; head -n 12 gpurun_out/kokoro_kernel_stats_v5.txt
