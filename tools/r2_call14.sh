#!/bin/bash
# round 2 call 14: flash_attn16 after the VALU cuts (tests + Whisper bench twice for variance)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "attention" > $O/t_fa16.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for i in 1 2; do
timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper_$i.json 2> $O/bench_whisper.err; echo "whisper rc=$?" >> $O/rc.txt
done
tail -3 $O/t_fa16.log; cat $O/rc.txt; for i in 1 2; do python -c "
import json
d=json.loads(open('$O/bench_whisper_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['split_ms'], d['attention_roofline']['achieved'], d['attention_roofline']['ms_per_launch'])"; done
