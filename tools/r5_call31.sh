#!/bin/bash
# round 5 call 31: SNAC encoder with LocalMHA (and preprocess with the window in its least common multiple), the SNAC decode file again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_codec_encode_gpu.py tests/test_snac_gpu.py -q -m gpu -k "snac or mha" -s > $O/pytest_snac_mha.txt 2>&1; echo "pytest rc=$?" >> $R
cat $R; tail -3 $O/pytest_snac_mha.txt | cut -c1-250; grep -E "^snac encoder with|^(FAILED|ERROR)|^E " $O/pytest_snac_mha.txt | head -12 | cut -c1-300
