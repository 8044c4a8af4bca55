#!/bin/bash
# round 6 call 3: (a) what binds conv_ws4 precision 5 -- the same launches with the HBM traffic / the data activity removed, power and clock alongside;
# (b) the SLP / packed-fp32 statistics defect of round 5: which operand form breaks it (variants of tools/build_variants.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python tools/conv_energy_probe.py --batch 64 --seconds 1.5 --out $O/conv_energy_probe.txt > $O/conv_energy_probe.log 2>&1; echo "energy probe rc=$?" >> $R
for v in 1 2 3 4; do
  echo "== variant $v" >> $O/slp_variants.txt
  MI355_LIB_PATH=$GRAFT_REPO_ROOT/mlx_audio_amd/lib/variants/libmi355audio_v$v.so timeout 200 python tools/diag_conv_stats.py --reps 3 >> $O/slp_variants.txt 2>&1; echo "slp v$v rc=$?" >> $R
done
cat $R; cat $O/conv_energy_probe.txt; grep -v amdgpu.ids $O/slp_variants.txt | cut -c1-220
