#!/bin/bash
# round 3 call 19: matrix pipe and HBM stream side by side in one kernel (power / clock budget probe)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 tools/bin/mfma_hbm_mix 200 > $O/mfma_hbm_mix.jsonl 2> $O/mfma_hbm_mix.err; echo "mix rc=$?" > $O/rc.txt
cat $O/rc.txt; cat $O/mfma_hbm_mix.jsonl; tail -3 $O/mfma_hbm_mix.err
