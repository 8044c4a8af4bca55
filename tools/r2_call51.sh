#!/bin/bash
# round 2 call 51: detect_language with the reference's (tokens, probability dicts) contract
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 40 python -m pytest tests/test_whisper_gpu.py -q -m gpu -x -k "detect_language or model_surface" > $O/t_whisper51.log 2>&1; echo "rc=$?"
tail -15 $O/t_whisper51.log
