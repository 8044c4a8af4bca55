#!/bin/bash
# round 3 call 39: where a step of the 256-wide LSTM recurrence goes: ablations (no recurrent product / no transcendentals / neither), kernel time from rocprofv3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
for v in 0 1 2 3; do
  MI355_LSTM_ABL=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-latency --no-roofline > $O/prof_l$v.log 2>&1
  DB=$(find $O/prof_l$v -name "*results.db" | head -1)
  echo "ABL $v: $(python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 | grep lstm | cut -c1-120)" >> $O/lstm_ablation.txt
  rm -rf $O/prof_l$v
done
cat $O/lstm_ablation.txt
