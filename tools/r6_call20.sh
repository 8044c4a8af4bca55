#!/bin/bash
# round 6 call 20: style projection as one item, shallow launches on ws4, polyphase interior epilogue: parity + contract line with the per-shape table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_kokoro_gpu.py tests/test_kitten_gpu.py -x -q > $O/pytest_call20.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 600 python bench.py --shape-table $O/shape_table_b64.txt --no-cpu-baseline --no-latency --no-secondary-precision > $O/bench_shape.json 2> $O/bench_shape.err; echo "bench rc=$?" >> $R
cat $R; tail -4 $O/pytest_call20.txt | cut -c1-200; cut -c1-330 $O/bench_shape.json; echo; head -50 $O/shape_table_b64.txt
