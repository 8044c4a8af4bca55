#!/bin/bash
# round 3 call 1: first run of gemm_rows.hip (9..64 rows per decode step): kernel parity, real-width stack parity, launch periods, Qwen3 line at 8 / 64 utterances
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "matrix_pipe" > $O/t_rows.log 2>&1; echo "rows rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py -q -m gpu -k "real_widths or prefill_and_decode" > $O/t_real.log 2>&1; echo "real rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_gemv.py --rows 64 --tag rows64 > $O/gemv_rows64.txt 2>&1
timeout 300 python tools/bench_gemv.py --rows 16 --tag rows16 > $O/gemv_rows16.txt 2>&1
timeout 300 python tools/bench_gemv.py --tag rows8 > $O/gemv_rows8.txt 2>&1
timeout 600 python tools/bench_qwen3.py --batch 64 --frames 24 --steps 1 --no-cpu-baseline > $O/qwen3_b64.json 2> $O/qwen3_b64.err; echo "q64 rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_qwen3.py --batch 8 --frames 24 --steps 1 --no-cpu-baseline > $O/qwen3_b8.json 2> $O/qwen3_b8.err; echo "q8 rc=$?" >> $O/rc.txt
tail -5 $O/t_rows.log; tail -5 $O/t_real.log; cat $O/rc.txt; grep -v "^{" $O/gemv_rows64.txt | head -16; head -c 600 $O/qwen3_b64.json; echo; head -c 400 $O/qwen3_b8.json; tail -3 $O/qwen3_b64.err
