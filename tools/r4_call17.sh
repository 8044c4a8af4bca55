#!/bin/bash
# round 4 call 17: which change breaks test_kokoro_batch_equals_single (A/B knobs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
T="tests/test_kokoro_gpu.py::test_kokoro_batch_equals_single"
timeout 600 python -m pytest $T -q -x > $O/t_default.txt 2>&1; echo "default rc=$?" > $O/rc.txt
MI355_ATTN_ONE_WAVE=0 timeout 600 python -m pytest $T -q -x > $O/t_4wave.txt 2>&1; echo "4wave rc=$?" >> $O/rc.txt
MI355_CONV_SPLIT_MINSTEPS=4 timeout 600 python -m pytest $T -q -x > $O/t_ms4.txt 2>&1; echo "minsteps4 rc=$?" >> $O/rc.txt
MI355_CONV_FINISH_OLD=1 timeout 600 python -m pytest $T -q -x > $O/t_oldfinish.txt 2>&1; echo "oldfinish rc=$?" >> $O/rc.txt
MI355_CONV_SPLIT=0 timeout 600 python -m pytest $T -q -x > $O/t_nosplit.txt 2>&1; echo "nosplit rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -h "AssertionError" $O/t_*.txt | head
