#!/bin/bash
# round-1 call 24: resident-x GEMV variant per decode shape (launch-period microbench), A/B against the chunked kernel
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
MI355_GEMV_RESIDENT=1 timeout 200 python tools/bench_gemv.py --tag resident --iters 200 > gpurun_out/gemv_resident.txt 2>&1; echo "rc=$?"
timeout 200 python tools/bench_gemv.py --tag chunked --iters 200 > gpurun_out/gemv_chunked.txt 2>&1; echo "rc=$?"
paste -d'\n' <(grep "us " gpurun_out/gemv_resident.txt | grep -v "^{") <(grep "us " gpurun_out/gemv_chunked.txt | grep -v "^{") | grep "M=8"
