#!/bin/bash
# round 2 call 52: the Qwen3 GPU tests that use the oracle's generate, after its next-input rule became a method
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 32 python -m pytest tests/test_codec_lm_gpu.py tests/test_tts_model_protocol_gpu.py -q -m gpu -x -k "qwen3" > $O/t_qwen52.log 2>&1; echo "rc=$?"
tail -6 $O/t_qwen52.log
