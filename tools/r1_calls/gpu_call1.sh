#!/bin/bash
# one gpurun call: WS-kernel tests first (bounded), then the full suite, the tile A/B, bench + rocprof stats
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 420 python -m pytest tests/test_kernels_gpu.py -q -k "8128128" --tb=short -p no:cacheprovider > gpurun_out/ws_tests.log 2>&1
rc=$?
echo "ws_tests rc=$rc" | tee gpurun_out/ws_rc.txt
if [ $rc -ne 0 ]; then export MI355_CONV_NO_WS=1; fi
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/ws_rc.txt
timeout 600 python tools/bench_conv.py --out gpurun_out/conv_ab.txt > gpurun_out/conv_ab.log 2>&1
echo "conv_ab rc=$?" | tee -a gpurun_out/ws_rc.txt
timeout 900 python bench.py --shape-table gpurun_out/shape_table.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" | tee -a gpurun_out/ws_rc.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/ws_rc.txt"
cd "$GRAFT_REPO_ROOT"
ls -la gpurun_out/prof | head
tail -5 gpurun_out/ws_tests.log; tail -15 gpurun_out/pytest.log; cat gpurun_out/conv_ab.txt; cat gpurun_out/bench.json
