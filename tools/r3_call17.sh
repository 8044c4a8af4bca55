#!/bin/bash
# round 3 call 17: (1) the resblock triple side by side on three streams vs one after the other; (2) the complete GPU suite with the split-K path in
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
echo skipped-concurrent
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -8 $O/pytest_gpu_full.txt
