#!/bin/bash
# round 3 call 30: per-kernel time of the Whisper step at 64 windows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_w -o p -- python $GRAFT_REPO_ROOT/tools/bench_whisper.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_w.log 2>&1
DB=$(find $O/prof_w -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 --by-grid > $O/kstats_whisper_b64.txt 2>&1
rm -rf $O/prof_w
cd $GRAFT_REPO_ROOT
head -30 $O/kstats_whisper_b64.txt | cut -c1-200
