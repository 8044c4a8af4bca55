#!/bin/bash
# call 36: vectorised fake-quant kernels: KittenTTS tests + bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kitten_gpu.py -q > gpurun_out/r2_kitten_call36.txt 2>&1; echo "kitten rc=$?" > gpurun_out/rc.txt
timeout 600 python bench.py --config kitten > gpurun_out/r2_bench_kitten_call36.json 2> gpurun_out/r2_bench_kitten_call36.err; echo "bench rc=$?" >> gpurun_out/rc.txt
tail -3 gpurun_out/r2_kitten_call36.txt; cat gpurun_out/rc.txt; head -c 900 gpurun_out/r2_bench_kitten_call36.json
