#!/bin/bash
# round 3 call 34: the contract command exactly as the driver runs it (defaults: PMC passes + cpu baseline + latency leg), wall time; batch-only kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
T0=$(date +%s)
timeout 1500 python bench.py > $O/bench_contract.json 2> $O/bench_contract.err; echo "contract rc=$? wall=$(( $(date +%s) - T0 ))s" > $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_contract.json")); r=d["roofline"]
print("value", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"], "ratio", r["traffic_over_algorithmic"], "lat", d["latency_b1"]["ms"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print(r["traffic_note"][:200])
PY
head -12 $O/kstats_b64.txt | cut -c1-170; tail -1 $O/prof_k.log | cut -c1-300
