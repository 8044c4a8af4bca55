#!/bin/bash
# experiment builds of the library for the SLP / packed-fp32 statistics defect (round 5): conv_ws4.hip (precision-2 instantiations) recompiled WITH the SLP
# vectoriser (-O3 default) and -DMI355_SLP_PROBE=n, linked with the production objects into mlx_audio_amd/lib/variants/libmi355audio_v<n>.so (select with
# MI355_LIB_PATH).  n = 1: the production source under SLP (the defect), 4: the production flags (-fno-slp-vectorize: exact).  (Variants 2 -- no SGPR operand
# in the packed epilogue arithmetic -- and 3 -- a 32-state pad + full drain in front of the epilogue -- were source branches under -DMI355_SLP_PROBE=n in
# conv_common.h; both still failed (profiles/r6_slp_hazard_variants_call3.txt) and the branches were removed from the production header at the end of round 6:
# git show c7f6d22:mlx_audio_amd/csrc/conv_common.h has them.)
set -e
cd "$(dirname "$0")/.."
L=mlx_audio_amd/lib; mkdir -p $L/variants
OBJS=$(ls $L/obj/*.o | grep -v "/conv_ws4.o")
for v in 1 4; do
  ( if [ $v = 4 ]; then F="-fno-slp-vectorize"; else F=""; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -x hip -c mlx_audio_amd/csrc/conv_ws4.hip -o $L/variants/conv_ws4_v$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/libmi355audio_v$v.so $OBJS $L/variants/conv_ws4_v$v.o && rm $L/variants/conv_ws4_v$v.o ) &
done
wait
ls -la $L/variants
