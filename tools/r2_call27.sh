#!/bin/bash
# round 2 call 27: LSTM with two gate rows per thread on v_dot2c (parity + time), shared instance-norm statistics
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py tests/test_edge_cases_gpu.py -q -m gpu > $O/t_lstm.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_k_lstm2.json 2> $O/bk1.err
MI355_LSTM_FMA=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_k_fma.json 2> $O/bk2.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-roofline > $O/prof_k.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_k -name "*results.db" | head -1) 3 > $O/kstats_kokoro.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
tail -5 $O/t_lstm.log; cat $O/rc.txt; for f in bench_k_lstm2 bench_k_fma; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done; grep -n "lstm\|instnorm" $O/kstats_kokoro.txt
