#!/bin/bash
# round 6 call 22: the full GPU suite + smoke() on the current build (tightened re-synchronisation thresholds, split activations, producer fast paths, polyphase interior epilogue)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $R
cat $R; tail -14 $O/pytest_full.txt | cut -c1-250; tail -3 $O/smoke.txt | cut -c1-300
