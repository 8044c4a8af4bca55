#!/bin/bash
# round 5 call 7: the whole GPU suite on the -fno-slp-vectorize build + the default bench line (with the in-run PMC passes) + LSTM / dsp lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
timeout 300 python bench.py --config dsp > $O/bench_dsp.json 2>> $O/bench_default.err; echo "bench dsp rc=$?" >> $R
cat $R; tail -5 $O/pytest_full.txt | cut -c1-300; cut -c1-3000 $O/bench_default.json; tail -3 $O/bench_default.err | cut -c1-300; cut -c1-400 $O/bench_dsp.json
