#!/bin/bash
# round 2 call 21: 16-bit KV caches in the stacks (tests), CSM with bf16 caches, fused-attention test tightened
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_tts_model_protocol_gpu.py tests/test_whisper_gpu.py -q -m gpu > $O/t_kv16s.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python tools/bench_csm.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_csm.json 2> $O/bench_csm.err; echo "csm rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_csm.py --steps 3 --warmup 1 --no-cpu-baseline --kv16 > $O/bench_csm_kv16.json 2> $O/bench_csm_kv16.err; echo "csm kv16 rc=$?" >> $O/rc.txt
tail -15 $O/t_kv16s.log; cat $O/rc.txt; for f in bench_csm bench_csm_kv16; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_frame'])"; done
