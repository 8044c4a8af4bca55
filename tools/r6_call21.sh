#!/bin/bash
# round 6 call 21: same-box A/B of the contract line: the tree at d9706c5 (before the producer / dispatch work of this part of the round) vs this tree, interleaved twice
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; : > $O/ab.txt
F="--no-cpu-baseline --no-latency --no-secondary-precision --no-roofline --no-batch-check --steps 20 --warmup 3"
for i in 1 2; do
  (cd _ab_old && timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['value'], d['ms_per_step'])") >> $O/ab.txt
  timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
(cd _ab_old && timeout 400 python bench.py --config whisper 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('whisper old', d['value'], d['ms_per_step'], d['split_ms'])") >> $O/ab.txt
timeout 400 python bench.py --config whisper 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('whisper new', d['value'], d['ms_per_step'], d['split_ms'])" >> $O/ab.txt
cat $O/ab.txt
