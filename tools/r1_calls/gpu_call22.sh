#!/bin/bash
# round-1 call 22: SNAC decode (next row f.2) against the oracle; the kernels it shares with other engines (dwconv: Qwen3 codec / Mimi) re-run
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 400 python -m pytest tests/test_snac_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/t_snac22.log 2>&1
echo "snac rc=$?" | tee -a $R
timeout 400 python -m pytest tests/test_qwen3_codec_gpu.py tests/test_mimi_gpu.py tests/test_dac_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_dw22.log 2>&1
echo "dwconv users rc=$?" | tee -a $R
cat $R; tail -n 60 gpurun_out/t_snac22.log | cut -c1-400; tail -n 5 gpurun_out/t_dw22.log
