#!/bin/bash
# round-1 call 19: Vocos + DAC parity, fbank, adaptive XCD run length check
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_vocos_gpu.py tests/test_dac_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/t_codec19.log 2>&1
echo "vocos+dac tests rc=$?" | tee -a $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "fbank or conv" -p no:cacheprovider > gpurun_out/t_k19.log 2>&1
echo "kernels subset rc=$?" | tee -a $R
timeout 300 python tools/bench_conv.py --batch 32 --rounds 5 --out gpurun_out/conv_tile_ab_b32_v5.txt > gpurun_out/bench_conv19.log 2>&1
echo "bench_conv rc=$?" | tee -a $R
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench19_b64.json 2> gpurun_out/bench19_b64.err
echo "bench b64 rc=$?" | tee -a $R
cat $R; grep -v "^$" gpurun_out/t_codec19.log | tail -n 40 | cut -c1-400; tail -n 5 gpurun_out/t_k19.log; grep "ws_regB" gpurun_out/conv_tile_ab_b32_v5.txt | cut -c1-120
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench19_b64.json").read().strip().splitlines()[-1])
print("b64", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
