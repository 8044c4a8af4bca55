#!/bin/bash
# round 2 call 13: 16-bit MFMA flash attention (tests + Whisper bench)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py -q -m gpu > $O/t_fa16.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "whisper rc=$?" >> $O/rc.txt
tail -12 $O/t_fa16.log; cat $O/rc.txt; python -c "
import json
d=json.loads(open('$O/bench_whisper.json').read().strip().splitlines()[-1]); print(d['value'], d['split_ms'], d['attention_roofline'])"
