#!/bin/bash
# round 4 call 9: unswizzled padded LDS planes of precision 5, batch-shaped adain_from_partials; p2 / p5 kernel traces of the same box (the control for the adain regression)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_conv_mx_gpu.py tests/test_kernels_gpu.py tests/test_kokoro_gpu.py -q -x -k "precision5 or adain or instnorm or batch_equals or teacher" > $O/pytest_a.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/bench_conv.py --prec-ab --batch 32 --out $O/conv_prec_ab_b32.txt > /dev/null 2> $O/conv_prec_ab.err; echo "prec_ab rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
for p in 5 2; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_p$p -o p -- python $GRAFT_REPO_ROOT/bench.py --precision $p --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_p$p.log 2>&1
  DB=$(find $O/prof_p$p -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64_p$p.txt 2>&1
  rm -rf $O/prof_p$p
done
MI355_ADAIN_CPW=4 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_cpw4 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_cpw4.log 2>&1
DB=$(find $O/prof_cpw4 -name "*results.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64_cpw4.txt 2>&1; rm -rf $O/prof_cpw4
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -3 $O/pytest_a.txt; grep "p5_\|p2_" $O/conv_prec_ab_b32.txt | awk '{print $1,$2,$3,$4,$7,$8,$11}'
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_p5.json")); r=d["roofline"]
print("p5", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3))
PY
for f in p5 p2 cpw4; do echo "== $f"; head -1 $O/kstats_b64_$f.txt; grep "adain_from\|conv_ws4_kernel<5, 2\|conv_ws4_kernel<2, 2, 0, false, false, 0, 128" $O/kstats_b64_$f.txt | cut -c1-130; done
