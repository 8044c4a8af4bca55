#!/bin/bash
# round 2 call 49: the CSM protocol test after the context / voice-match change of generate()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 60 python -m pytest tests/test_tts_model_protocol_gpu.py -q -m gpu -k csm > $O/t_csm49.log 2>&1; echo "rc=$?"
tail -5 $O/t_csm49.log
