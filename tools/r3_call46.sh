#!/bin/bash
# round 3 call 46: Qwen3-TTS speech tokenizer ENCODER on the Mimi engine (HF checkpoint -> sanitize -> load_weights -> encode) against the reference run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_reference_fixtures_gpu.py -q -m gpu -k "tokenizer_encoder or mimi" > $O/t_tok.log 2>&1; echo "tok rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -25 $O/t_tok.log
