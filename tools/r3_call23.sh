#!/bin/bash
# round 3 call 23: Mimi encoder (SEANet encoder + transformer + edge-padded downsample + mi355_rvq_encode): parity vs the reference run and the oracle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_reference_fixtures_gpu.py tests/test_mimi_gpu.py -q -m gpu -k "mimi" > $O/t_mimi.log 2>&1; echo "mimi rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_tts_model_protocol_gpu.py tests/test_codec_lm_gpu.py -q -m gpu -k "csm or sesame" > $O/t_csm.log 2>&1; echo "csm rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -30 $O/t_mimi.log; tail -5 $O/t_csm.log
