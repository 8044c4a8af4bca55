#!/bin/bash
# round 3 call 36: validation of the round's last build: complete GPU suite, smoke(), the contract command, secondary lines, batch-only kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1500 python bench.py > $O/bench_contract.json 2> $O/bench_contract.err; echo "contract rc=$?" >> $O/rc.txt
for c in whisper qwen3 csm kitten; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -4 $O/pytest_gpu_full.txt; tail -1 $O/smoke.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_contract.json")); r=d["roofline"]
print("contract", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "traffic ratio", round(r["traffic_over_algorithmic"],4), "lat", round(d["latency_b1"]["ms"],3), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
for n in ("whisper","qwen3","csm","kitten"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "roofline", round((d.get("roofline") or {}).get("frac",0),4))
    except Exception as e: print(n, "ERR", e)
PY
head -8 $O/kstats_b64.txt | cut -c1-150
