#!/bin/bash
# round 6 call 23: is the first process on a fresh box slower because the chip is cold?  the contract command with 5 and then 60 warm-up steps, then 5 again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; : > $O/warm.txt
F="--no-cpu-baseline --no-latency --no-secondary-precision --no-roofline --no-batch-check --steps 20"
for w in 5 60 5 5; do
  timeout 300 python bench.py $F --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup', d['warmup'], d['value'], d['ms_per_step'])" >> $O/warm.txt
done
cat $O/warm.txt
