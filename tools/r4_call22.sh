#!/bin/bash
# round 4 call 22: kernel traces (by launch shape) of the secondary lines on the round's final build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_c.log 2>&1
DB=$(find $O/prof_c -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 --by-grid > $O/kstats_csm.txt 2>&1
rm -rf $O/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --no-cpu-baseline --batch 64 --steps 1 --warmup 1 > $O/prof_q.log 2>&1
DB=$(find $O/prof_q -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_qwen3_b64.txt 2>&1
rm -rf $O/prof_q
tail -3 $O/prof_c.log | cut -c1-300; tail -3 $O/prof_q.log | cut -c1-300
head -22 $O/kstats_csm.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
head -26 $O/kstats_qwen3_b64.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
