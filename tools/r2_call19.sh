#!/bin/bash
# round 2 call 19: deferred final norms (CSM depth decoder, Qwen3 code predictor), edge-case tests again, Whisper step kernel timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_edge_cases_gpu.py tests/test_codec_lm_gpu.py tests/test_tts_model_protocol_gpu.py tests/test_lm_kernels_gpu.py -q -m gpu > $O/t_defer.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for c in csm qwen3; do
timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w -o p -- python $GRAFT_REPO_ROOT/bench.py --config whisper --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_w.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_w -name "*results.db" | head -1) 2 --by-grid > $O/kstats_whisper_bygrid.txt 2>&1
rm -rf $O/prof_w
cd $GRAFT_REPO_ROOT
tail -6 $O/t_defer.log; cat $O/rc.txt; for f in bench_csm bench_qwen3; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_frame'])"; done; grep -n "greedy" $O/kstats_whisper_bygrid.txt
