#!/bin/bash
# round 3 call 50: every GPU test that goes through the files touched after the last full run (qwen3_tts.py routing / batch, talker.py, ops.py additions)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 160 python -m pytest tests/test_tts_model_protocol_gpu.py tests/test_qwen3_clone_gpu.py -q -m gpu -x > $O/t_proto_all.log 2>&1; echo "proto rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -12 $O/t_proto_all.log
