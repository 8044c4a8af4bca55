#!/bin/bash
# round 5 call 35: DAC compress / decompress on the device against the reference run
# default-constructed engines changed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_codec_encode_gpu.py tests/test_snac_gpu.py tests/test_dac_gpu.py tests/test_encodec_gpu.py -q -m gpu -k "compress" > $O/pytest_ref_tests.txt 2>&1; echo "pytest rc=$?" >> $R
cat $R; tail -3 $O/pytest_ref_tests.txt | cut -c1-250; grep -E "^(FAILED|ERROR)|^E " $O/pytest_ref_tests.txt | head -12 | cut -c1-300
