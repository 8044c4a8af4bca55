#!/bin/bash
# round 5 call 5: which change to the interior epilogue of the ws4 kernel removes the random M2 errors of its fused statistics?
# v1 = generic epilogue for every tile; v2 = full waits around the cross-half exchange; v3 = exchange by v_permlane32_swap; v4 = no SLP vectorisation
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
{
echo "== production"; timeout 120 python tools/diag_conv_stats.py --L 256 --reps 3 | grep -v "^   st"
for v in 1 2 3 4; do
  echo "== variant $v"; MI355_LIB_PATH=$GRAFT_REPO_ROOT/mlx_audio_amd/lib/variants/libmi355audio_v$v.so timeout 120 python tools/diag_conv_stats.py --L 256 --reps 3 | grep -v "^   st"
done
} > $O/diag_conv_stats_variants.txt 2>&1
grep -v amdgpu.ids $O/diag_conv_stats_variants.txt | cut -c1-260
