#!/bin/bash
# round 2 call 50: the GPU tests that go through KokoroPipeline, after its chunking was rewritten to mirror the reference
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 55 python -m pytest tests/test_api_gpu.py -q -m gpu -x -k "kokoro_model_protocol" > $O/t_api50.log 2>&1; echo "rc=$?"
tail -4 $O/t_api50.log
