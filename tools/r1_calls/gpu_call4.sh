#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $R
timeout 400 python tools/bench_conv.py --ablate --out gpurun_out/conv_ablate.txt > gpurun_out/conv_ab.log 2>&1
echo "ablate rc=$?" | tee -a $R
cat $R; tail -15 gpurun_out/pytest.log; cat gpurun_out/conv_ablate.txt
