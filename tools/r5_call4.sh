#!/bin/bash
# round 5 call 4: the fused statistics of ONE ws4 conv launch on identical rows (first row-dependent launch of call 3)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
{
echo "== auto B=64 L=264 C=512 k=3"; timeout 120 python tools/diag_conv_stats.py
echo "== auto, no lens"; timeout 120 python tools/diag_conv_stats.py --nolens --reps 2
echo "== ws4 one workgroup per tile (feature bit 3)"; timeout 120 python tools/diag_conv_stats.py --tile 86128128 --reps 2
echo "== ws4 explicit"; timeout 120 python tools/diag_conv_stats.py --tile 6128128 --reps 2
echo "== 4-wave 128x128"; timeout 120 python tools/diag_conv_stats.py --tile 128128 --reps 2
echo "== auto B=64 L=256 (no ragged tile)"; timeout 120 python tools/diag_conv_stats.py --L 256 --reps 2
echo "== auto B=64 L=330 (ragged tile > 64 rows)"; timeout 120 python tools/diag_conv_stats.py --L 330 --reps 2
echo "== auto B=4 L=5280 C=256 k=7 p5"; timeout 120 python tools/diag_conv_stats.py --batch 4 --L 5280 --C 256 --k 7 --precision 5 --reps 2
echo "== serialized kernels"; AMD_SERIALIZE_KERNEL=3 timeout 120 python tools/diag_conv_stats.py --reps 2
} > $O/diag_conv_stats.txt 2>&1
grep -v amdgpu.ids $O/diag_conv_stats.txt | cut -c1-260
