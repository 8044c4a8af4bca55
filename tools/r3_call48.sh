#!/bin/bash
# round 3 call 48: the cloning end-to-end test again (its fixture loaded the tiny codec decoder through the shape heuristic of sanitize: fixed in the test)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 170 python -m pytest tests/test_qwen3_clone_gpu.py -q -m gpu -s > $O/t_clone.log 2>&1; echo "clone rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -40 $O/t_clone.log
