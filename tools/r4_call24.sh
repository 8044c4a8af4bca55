#!/bin/bash
# round 4 call 24: the driver's round-end sequence on the final commit: full GPU suite, smoke(), the default bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_final.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1500 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_final.txt | cut -c1-200; tail -1 $O/smoke.txt; cat $O/bench_final.json | cut -c1-400
