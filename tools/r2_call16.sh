#!/bin/bash
# round 2 call 16: flash_attn16 with the accumulator started at -m; key-split A/B for the 16-bit cross-attention
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py -q -m gpu > $O/t_fa16.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for ns in 0 2 3; do
MI355_ATTN_NSPLIT=$ns timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper_ns$ns.json 2> $O/bench_whisper.err; echo "whisper ns=$ns rc=$?" >> $O/rc.txt
done
tail -3 $O/t_fa16.log; cat $O/rc.txt; for ns in 0 2 3; do python -c "
import json
d=json.loads(open('$O/bench_whisper_ns$ns.json').read().strip().splitlines()[-1]); print($ns, d['value'], d['split_ms'], d['attention_roofline']['achieved'], d['attention_roofline']['ms_per_launch'])"; done
