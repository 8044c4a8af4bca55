#!/bin/bash
# round 3 call 28: the complete GPU suite + smoke() with everything of calls 14-27 in
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -12 $O/pytest_gpu_full.txt; tail -2 $O/smoke.log
