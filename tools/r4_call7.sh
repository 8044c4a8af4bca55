#!/bin/bash
# round 4 call 7: the complete GPU suite after the round's changes (precision 5, one-row GEMV, margin rule with re-synchronisation, streaming decoders, advisor fixes)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
MI355_MARGIN_REPORT=$O/margin_report.txt timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -25 $O/pytest_gpu_full.txt | cut -c1-220; tail -1 $O/smoke.log
