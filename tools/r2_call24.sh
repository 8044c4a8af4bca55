#!/bin/bash
# round 2 call 24: K > 2048 through the staged matrix-pipe GEMV's chunk loop (A/B) + numerics
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/bench_gemv.py --iters 300 --tag default > $O/gemv_default.txt 2>&1
MI355_GEMV_MFMA_CHUNKED=1 timeout 300 python tools/bench_gemv.py --iters 300 --tag chunked > $O/gemv_chunked.txt 2>&1
MI355_GEMV_MFMA_CHUNKED=1 timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "gemv" > $O/t_chunk.log 2>&1; echo "tests rc=$?" > $O/rc.txt
MI355_GEMV_MFMA_CHUNKED=1 timeout 600 python bench.py --config qwen3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_qwen3_chunked.json 2> $O/bq.err
MI355_GEMV_MFMA_CHUNKED=1 timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper_chunked.json 2> $O/bw.err
grep -n "mlp2\|down" $O/gemv_default.txt $O/gemv_chunked.txt; tail -3 $O/t_chunk.log; python -c "
import json
d=json.loads(open('$O/bench_qwen3_chunked.json').read().strip().splitlines()[-1]); print('qwen3 chunked', d['value'], d['ms_per_frame'])
d=json.loads(open('$O/bench_whisper_chunked.json').read().strip().splitlines()[-1]); print('whisper chunked', d['value'], d['split_ms'])"
