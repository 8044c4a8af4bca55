#!/bin/bash
# round 5 call 17: HBM bytes per generated frame of the decode chains by the TCC counters ("keep the depth decoder resident in the
# Infinity Cache across its 31 steps (measure FETCH_SIZE to prove it)"), and a last sanity run of the kernels touched since the validation call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm or conv_gemm_plain or silent or identical" > $O/pytest_touch.txt 2>&1; echo "pytest touched kernels rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_csm -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --steps 1 --warmup 0 --frames 32 > $O/pmc_csm.log 2>&1; echo "pmc csm rc=$?" >> $R
DB=$(find $O/pmc_csm -name "*_results.db" | head -1); (cd $GRAFT_REPO_ROOT/tools && python pmc_total.py "$DB" FETCH_SIZE 32) > $O/pmc_fetch_csm.txt 2>&1; rm -rf $O/pmc_csm
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_q3 -o p -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --no-cpu-baseline --steps 1 --warmup 0 --frames 16 > $O/pmc_q3.log 2>&1; echo "pmc qwen3 rc=$?" >> $R
DB=$(find $O/pmc_q3 -name "*_results.db" | head -1); (cd $GRAFT_REPO_ROOT/tools && python pmc_total.py "$DB" FETCH_SIZE 16) > $O/pmc_fetch_qwen3.txt 2>&1; rm -rf $O/pmc_q3
cd $GRAFT_REPO_ROOT
cat $R; tail -2 $O/pytest_touch.txt | cut -c1-200; cat $O/pmc_fetch_csm.txt | cut -c1-200; tail -2 $O/pmc_csm.log | cut -c1-400; cat $O/pmc_fetch_qwen3.txt | cut -c1-200; tail -2 $O/pmc_q3.log | cut -c1-400
