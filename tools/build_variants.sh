#!/bin/bash
# experiment builds of the library: conv_ws4.hip (precision-2 instantiations) recompiled with -DMI355_VARIANT=n, linked with the production objects
# into mlx_audio_amd/lib/variants/libmi355audio_v<n>.so (select with MI355_LIB_PATH).  Variant 4 = the production source with -fno-slp-vectorize.
set -e
cd "$(dirname "$0")/.."
L=mlx_audio_amd/lib; mkdir -p $L/variants
OBJS=$(ls $L/obj/*.o | grep -v "/conv_ws4.o")
for v in 1 2 3 4; do
  ( if [ $v = 4 ]; then F="-fno-slp-vectorize"; else F="-DMI355_VARIANT=$v"; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -x hip -c mlx_audio_amd/csrc/conv_ws4.hip -o $L/variants/conv_ws4_v$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/libmi355audio_v$v.so $OBJS $L/variants/conv_ws4_v$v.o && rm $L/variants/conv_ws4_v$v.o ) &
done
wait
ls -la $L/variants
