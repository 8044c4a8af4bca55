#!/bin/bash
# round 3 call 24: CSM audio context through the Mimi encoder; the Mimi / CSM tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_tts_model_protocol_gpu.py -q -m gpu -k "csm" > $O/t_csm.log 2>&1; echo "csm rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -30 $O/t_csm.log
