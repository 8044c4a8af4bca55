#!/bin/bash
# round 2 call 8: new parity tests (protocol, interpolate, whisper prompt/sampling/fixture), full suite with margin report
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MI355_MARGIN_REPORT=gpurun_out/margin_report.txt
timeout 900 python -m pytest tests/test_tts_model_protocol_gpu.py tests/test_interpolate_gpu.py tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py -q -m gpu -s > gpurun_out/t_new.log 2>&1; echo "new tests rc=$?" > gpurun_out/rc.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t_full.log 2>&1; echo "full rc=$?" >> gpurun_out/rc.txt
tail -5 gpurun_out/t_new.log; tail -8 gpurun_out/t_full.log; cat gpurun_out/rc.txt
