#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 420 python -m pytest tests/test_kernels_gpu.py -q -k "8128128 or 9128128" --tb=short -p no:cacheprovider > gpurun_out/ws_tests.log 2>&1
rc=$?; echo "ws_tests rc=$rc" | tee -a $R
if [ $rc -ne 0 ]; then export MI355_CONV_NO_WS=1; fi
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $R
timeout 400 python tools/bench_conv.py --out gpurun_out/conv_ab_rot.txt > gpurun_out/conv_ab.log 2>&1
echo "conv_ab rot rc=$?" | tee -a $R
MI355_CONV_NO_ROT=1 timeout 400 python tools/bench_conv.py --out gpurun_out/conv_ab_norot.txt >> gpurun_out/conv_ab.log 2>&1
echo "conv_ab norot rc=$?" | tee -a $R
timeout 400 python tools/bench_conv.py --precision 3 --out gpurun_out/conv_ab_rot_fp16.txt >> gpurun_out/conv_ab.log 2>&1
echo "conv_ab fp16 rc=$?" | tee -a $R
timeout 600 python bench.py --shape-table gpurun_out/shape_table.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" | tee -a $R
timeout 300 python bench.py --precision 3 --no-cpu-baseline --shape-table gpurun_out/shape_table_p3.txt > gpurun_out/bench_p3.json 2>> gpurun_out/bench.err
echo "bench p3 rc=$?" | tee -a $R
MI355_CONV_WS_VARIANT=8 timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_v8.json 2>> gpurun_out/bench.err
echo "bench v8 rc=$?" | tee -a $R
timeout 300 python bench.py --batch 1 --no-cpu-baseline > gpurun_out/bench_b1.json 2>> gpurun_out/bench.err
echo "bench b1 rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq" -o pmc -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --quick --rounds 2 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log" 2>&1
echo "pmc sq rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_tcc" -o pmc -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --quick --rounds 2 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_tcc.log" 2>&1
echo "pmc tcc rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch" -o pmc -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --quick --rounds 2 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_write" -o pmc -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --quick --rounds 2 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_write.log" 2>&1
echo "pmc write rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
cat $R; tail -4 gpurun_out/ws_tests.log; tail -25 gpurun_out/pytest.log; cat gpurun_out/bench.json gpurun_out/bench_p3.json gpurun_out/bench_v8.json gpurun_out/bench_b1.json
