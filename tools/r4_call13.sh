#!/bin/bash
# round 4 call 13: full GPU suite on the straight-line decode kernels (one-row GEMVs, row epilogue, decode attention, pinned rotary roundings) and the
# one-sweep AdaIN coefficients kernel with its loads in flight; Kokoro contract-like line + kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -6 $O/pytest_full.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_p5.json")); r=d["roofline"]
print("p5", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3))
PY
head -30 $O/kstats_b64.txt | cut -c1-150
