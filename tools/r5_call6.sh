#!/bin/bash
# round 5 call 6: the library built with -fno-slp-vectorize: identical-row diagnostics, the regression tests, Kokoro parity (incl. B = 64), bench (speed with / without SLP)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python tools/diag_batch_rows.py --precision 5 --batch 64 --repeat 2 > $O/diag_rows_p5_b64_noslp.txt 2>&1; echo "diag p5 b64 rc=$?" >> $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "identical_rows or instnorm or lstm or adain" > $O/pytest_stats.txt 2>&1; echo "pytest stats rc=$?" >> $R
timeout 900 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 600 python bench.py --no-pmc > $O/bench_default_noslp.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; grep -v amdgpu.ids $O/diag_rows_p5_b64_noslp.txt | cut -c1-200; tail -3 $O/pytest_stats.txt | cut -c1-300; grep -a "kokoro\|passed\|failed\|Error" $O/pytest_kokoro.txt | cut -c1-300 | tail -20
cut -c1-2500 $O/bench_default_noslp.json; tail -3 $O/bench_default.err | cut -c1-300
