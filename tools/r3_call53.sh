#!/bin/bash
# round 3 call 53: sustained rate of the 8-bit matrix shapes next to bf16 / fp16 on this box (random and zero operands): what a "lo pass on an 8-bit pipe" could buy
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 40 tools/bin/mfma_peak 200 > $O/mfma_peak8_random.jsonl 2>&1; echo "random rc=$?" > $O/rc.txt
MI355_MFMA_ZERO=1 timeout 40 tools/bin/mfma_peak 200 > $O/mfma_peak8_zero.jsonl 2>&1; echo "zero rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/mfma_peak8_random.jsonl $O/mfma_peak8_zero.jsonl
