#!/bin/bash
# round 4 call 18: where the batch-vs-single difference of split groups of >= 2 steps sits (tools/diag_batch_single.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/diag_batch_single.py > $O/diag_ms2.txt 2>&1; echo "ms2 rc=$?" > $O/rc.txt
MI355_CONV_SPLIT_MINSTEPS=4 timeout 600 python tools/diag_batch_single.py > $O/diag_ms4.txt 2>&1; echo "ms4 rc=$?" >> $O/rc.txt
cat $O/rc.txt; echo ---- ms2; tail -40 $O/diag_ms2.txt; echo ---- ms4; grep "^item" $O/diag_ms4.txt
