#!/bin/bash
# call 33: KittenTTS tests (all) + the KittenTTS secondary bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kitten_gpu.py -q -s > gpurun_out/r2_kitten_call33.txt 2>&1; echo "kitten rc=$?" > gpurun_out/rc.txt
timeout 900 python bench.py --config kitten > gpurun_out/r2_bench_kitten_call33.json 2> gpurun_out/r2_bench_kitten_call33.err; echo "bench rc=$?" >> gpurun_out/rc.txt
grep -n "HIP vs\|batch vs\|kitten\|passed\|failed\|Error" gpurun_out/r2_kitten_call33.txt | head -40; cat gpurun_out/r2_bench_kitten_call33.json; tail -5 gpurun_out/r2_bench_kitten_call33.err; cat gpurun_out/rc.txt
