#!/bin/bash
# round 3 call 2: ablation of gemm_rows_kernel at 64 rows on the talker / code-predictor shapes (what bounds it: x re-reads, conversion, LDS writes, MFMA, weights)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
: > $O/rows_ablate.txt
for dbg in 0 1 2 4 8 16 32 7 39 24 47; do
  MI355_GEMM_ROWS_DBG=$dbg timeout 120 python tools/bench_gemv.py --rows 64 --graph --iters 100 --only "talker" --tag dbg$dbg 2>&1 | grep -v "^{" | grep -v amdgpu >> $O/rows_ablate.txt
done
for w in 128 256; do
  MI355_GEMM_ROWS_WGS=$w timeout 120 python tools/bench_gemv.py --rows 64 --graph --iters 100 --only "talker" --tag wgs$w 2>&1 | grep -v "^{" | grep -v amdgpu >> $O/rows_ablate.txt
done
timeout 120 python tools/bench_gemv.py --rows 8 --graph --iters 100 --only "talker" --tag rows8g 2>&1 | grep -v "^{" | grep -v amdgpu >> $O/rows_ablate.txt
timeout 120 python tools/bench_gemv.py --rows 1 --graph --iters 100 --only "talker" --tag rows1g 2>&1 | grep -v "^{" | grep -v amdgpu >> $O/rows_ablate.txt
cat $O/rows_ablate.txt
