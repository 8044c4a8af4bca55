#!/bin/bash
# round 2 call 23: multi-workgroup Whisper decode-rules step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py tests/test_api_gpu.py -q -m gpu -k "whisper or greedy" > $O/t_ws.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "whisper rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w -o p -- python $GRAFT_REPO_ROOT/bench.py --config whisper --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_w.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_w -name "*results.db" | head -1) 2 --by-grid > $O/kstats_whisper_bygrid.txt 2>&1
rm -rf $O/prof_w
cd $GRAFT_REPO_ROOT
tail -5 $O/t_ws.log; cat $O/rc.txt; python -c "
import json
d=json.loads(open('$O/bench_whisper.json').read().strip().splitlines()[-1]); print(d['value'], d['split_ms'])"; grep -n "whisper_step\|greedy" $O/kstats_whisper_bygrid.txt
