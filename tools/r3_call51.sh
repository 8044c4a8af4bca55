#!/bin/bash
# round 3 call 51: batch_generate(stream=True) on the slot engine (plain and shared-reference batches)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 python -m pytest tests/test_qwen3_clone_gpu.py -q -m gpu -k "streams_chunks" > $O/t_stream.log 2>&1; echo "stream rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -30 $O/t_stream.log
