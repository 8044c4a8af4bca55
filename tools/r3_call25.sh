#!/bin/bash
# round 3 call 25: per-kernel time of the 64-utterance Kokoro step (rocprofv3 --kernel-trace --stats), current build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 --by-grid > $O/kstats_b64_bygrid.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
tail -1 $O/prof_k.log | cut -c1-400; head -36 $O/kstats_b64.txt | cut -c1-170
