#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bigvgan_gpu.py -q -s -x > gpurun_out/r2_bigvgan_call40.txt 2>&1; echo "rc=$?" > gpurun_out/rc.txt
grep -n "bigvgan resblock\|passed\|failed\|Error\|assert" gpurun_out/r2_bigvgan_call40.txt | head -20; tail -25 gpurun_out/r2_bigvgan_call40.txt; cat gpurun_out/rc.txt
