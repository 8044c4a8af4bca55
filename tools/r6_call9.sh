#!/bin/bash
# round 6 call 9: one decode frame of Qwen3-TTS-1.7B at 64 and at 8 utterances (the timed leg, not the one-sequence time-to-first-audio leg at the trace's end)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
cd /tmp && export TMPDIR=/tmp
for b in 64 8; do
  timeout 400 rocprofv3 --kernel-trace -d $O/prof_q_$b -o t -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --batch $b --frames 12 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/prof_q_$b.err; echo "trace qwen3 $b rc=$?" >> $R
  DB=$(find $O/prof_q_$b -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py "$DB" sample_kernel 40 --span=16 --from-start=272 > $O/timeline_qwen3_b$b.txt 2>&1
  rm -rf $O/prof_q_$b
done
cd $GRAFT_REPO_ROOT; cat $R; for b in 64 8; do echo "== b$b"; head -24 $O/timeline_qwen3_b$b.txt | cut -c1-170; done
