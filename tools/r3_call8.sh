#!/bin/bash
# round 3 call 8: Qwen3-TTS continuous batching on slot KV caches + everything touched since call 6 (ADVICE fixes, tightened bars, latency_b1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_tts_model_protocol_gpu.py -q -m gpu -x -k "qwen3" > $O/t_sess.log 2>&1; echo "sess rc=$?" > $O/rc.txt
tail -25 $O/t_sess.log; cat $O/rc.txt
