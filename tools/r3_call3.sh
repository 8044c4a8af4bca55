#!/bin/bash
# round 3 call 3: rows pipeline (rows_pipe.hip): kernel parity, real-width stack parity through the native runner, stage timings, Qwen3 line at 64 utterances
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "rows or tile_image" > $O/t_pipe.log 2>&1; echo "pipe rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py -q -m gpu -k "real_widths" > $O/t_real.log 2>&1; echo "real rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_gemv.py --rows 64 --pipe --iters 100 --tag pipe64 > $O/pipe_rows64.txt 2>&1
timeout 300 python tools/bench_gemv.py --rows 16 --pipe --iters 100 --tag pipe16 --only talker > $O/pipe_rows16.txt 2>&1
timeout 600 python tools/bench_qwen3.py --batch 64 --frames 24 --steps 1 --no-cpu-baseline > $O/qwen3_b64.json 2> $O/qwen3_b64.err; echo "q64 rc=$?" >> $O/rc.txt
tail -5 $O/t_pipe.log; tail -5 $O/t_real.log; cat $O/rc.txt; grep -v "^{" $O/pipe_rows64.txt | grep -v amdgpu; grep -v "^{" $O/pipe_rows16.txt | grep -v amdgpu; python -c "
import json
d=json.load(open('$O/qwen3_b64.json')); print('qwen3 b64', d['value'], d['split_ms'], d['ms_per_frame'], d['roofline']['frac'])"; tail -3 $O/qwen3_b64.err
