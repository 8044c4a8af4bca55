#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/debug_batch_extreme.py > $O/debug_extreme.txt 2>&1
timeout 600 python -m pytest tests/test_edge_cases_gpu.py tests/test_transformer_kernels_gpu.py -q -m gpu -k "whisper or greedy or empty" > $O/t_edge.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper.json 2> $O/bench_whisper.err
cat $O/debug_extreme.txt | head -80; tail -5 $O/t_edge.log; python -c "
import json
d=json.loads(open('$O/bench_whisper.json').read().strip().splitlines()[-1]); print(d['value'], d['split_ms'])"
