#!/bin/bash
# round 3 call 5: kernel trace of the Qwen3-TTS-1.7B frame loop at 64 utterances (where the 11.4 ms per frame go)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --batch 64 --frames 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_q.log 2>&1
DB=$(find $O/prof_q -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_qwen3_b64.txt 2>&1
rm -rf $O/prof_q
head -45 $O/kstats_qwen3_b64.txt; tail -2 $O/prof_q.log | head -c 600
