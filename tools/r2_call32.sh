#!/bin/bash
# call 32: KittenTTS tests + regression of what the engine changes touch
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kitten_gpu.py -x -q -s > gpurun_out/r2_kitten_call32.txt 2>&1; echo "kitten rc=$?" > gpurun_out/rc.txt
timeout 1200 python -m pytest tests/test_kokoro_gpu.py tests/test_kernels_gpu.py -q -x -k "kokoro or lstm or duration or sine or source" > gpurun_out/r2_kokoro_regress_call32.txt 2>&1; echo "regress rc=$?" >> gpurun_out/rc.txt
tail -30 gpurun_out/r2_kitten_call32.txt; tail -5 gpurun_out/r2_kokoro_regress_call32.txt; cat gpurun_out/rc.txt
