#!/bin/bash
# round 6 call 15: where the Whisper step goes, kernel by kernel (rocprofv3 --kernel-trace --stats of the Whisper line)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
MI355_WHISPER_SPLIT=0 timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_w -o w -- python $GRAFT_REPO_ROOT/tools/bench_whisper.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper_prof.json 2> $O/prof_w.err; echo "rc=$?"
DB=$(find $O/prof_w -name "*_results.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 4 > $O/kernel_stats_whisper.txt 2>/dev/null
rm -rf $O/prof_w
head -40 $O/kernel_stats_whisper.txt | cut -c1-200; tail -3 $O/prof_w.err
