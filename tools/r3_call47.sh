#!/bin/bash
# round 3 call 47: Qwen3-TTS voice cloning on the device: ECAPA row kernels, speaker encoder vs the reference run and at the published widths,
# Model x-vector / in-context prompts / generate / batch; plus the protocol test whose refusal assertion changed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 170 python -m pytest tests/test_qwen3_clone_gpu.py -q -m gpu -s > $O/t_clone.log 2>&1; echo "clone rc=$?" > $O/rc.txt
timeout 90 python -m pytest tests/test_tts_model_protocol_gpu.py -q -m gpu -k "qwen3_tts_load_model_and_generate" > $O/t_proto.log 2>&1; echo "proto rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -40 $O/t_clone.log; tail -5 $O/t_proto.log
