#!/bin/bash
# round 2 call 17: fused norm/rope/cache-store decode attention (Qwen3 stacks), edge-case tests, sharded-step host path; Qwen3 / Kokoro bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_edge_cases_gpu.py tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_tts_model_protocol_gpu.py -q -m gpu > $O/t_fuse.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config qwen3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_qwen3.json 2> $O/bench_qwen3.err; echo "qwen3 rc=$?" >> $O/rc.txt
MI355_ATTN_FUSE_ROPE=0 timeout 600 python bench.py --config qwen3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_qwen3_nofuse.json 2> $O/bench_qwen3.err; echo "qwen3 nofuse rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_kokoro3.json 2> $O/bench_kokoro3.err; echo "kokoro rc=$?" >> $O/rc.txt
tail -12 $O/t_fuse.log; cat $O/rc.txt; for f in bench_qwen3 bench_qwen3_nofuse; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_frame'])"; done; python -c "
import json
d=json.loads(open('$O/bench_kokoro3.json').read().strip().splitlines()[-1]); print('kokoro', d['value'], d['ms_per_step'])"
