#!/bin/bash
# round 5 call 28: edge cases of the codec encode sides (one hop, ragged lengths, one-sample SNAC input, EnCodec in a batch)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_codec_encode_gpu.py -q -m gpu -k "edge" -s > $O/pytest_encode_edge.txt 2>&1; echo "pytest rc=$?" >> $R
cat $R; tail -3 $O/pytest_encode_edge.txt | cut -c1-250; grep -E "^(FAILED|ERROR)|^E " $O/pytest_encode_edge.txt | head -12 | cut -c1-300
