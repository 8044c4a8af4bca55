#!/bin/bash
# call 35: the new reference-run GPU tests (sampler, log-mel) + the suites touched since call 34
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_fixtures_gpu.py tests/test_mimi_gpu.py tests/test_snac_gpu.py tests/test_lm_kernels_gpu.py -q -s > gpurun_out/r2_fixtures_call35.txt 2>&1; echo "rc=$?" > gpurun_out/rc.txt
grep -n "vs reference run\|passed\|failed\|Error" gpurun_out/r2_fixtures_call35.txt | head -30; cat gpurun_out/rc.txt
