#!/bin/bash
# round 3 call 54: operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4, by experiment
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 30 tools/bin/mfma_mx_probe > $O/mfma_mx_probe.jsonl 2>&1; echo "probe rc=$?" > $O/rc.txt
cat $O/rc.txt $O/mfma_mx_probe.jsonl
