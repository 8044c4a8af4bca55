#!/bin/bash
# round-2 call 3: persistent ws4 (cross-tile producer prefetch) vs one workgroup per tile vs ws3; timeline probe per tile
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --tb=short -p no:cacheprovider -k "conv" > gpurun_out/t_conv3.log 2>&1
echo "conv tests rc=$?" | tee -a $R
timeout 300 python tools/conv_timeline.py --batch 32 --out gpurun_out/conv_timeline3_b32.txt > /dev/null 2> gpurun_out/conv_timeline.err
echo "timeline rc=$?" | tee -a $R
timeout 400 python tools/bench_conv.py --batch 32 --out gpurun_out/conv_ab3_b32.txt > /dev/null 2> gpurun_out/conv_ab3_b32.err
echo "bench_conv rc=$?" | tee -a $R
timeout 600 python bench.py --no-cpu-baseline --shape-table gpurun_out/shape_table3_ws4.txt > gpurun_out/bench3.json 2> gpurun_out/bench3.err
echo "bench rc=$?" | tee -a $R
cat $R; tail -n 5 gpurun_out/t_conv3.log | cut -c1-250
cat gpurun_out/conv_timeline3_b32.txt; tail -n 5 gpurun_out/conv_timeline.err
cat gpurun_out/conv_ab3_b32.txt
cat gpurun_out/bench3.json | cut -c1-1500; tail -n 3 gpurun_out/bench3.err
