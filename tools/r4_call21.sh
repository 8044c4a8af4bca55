#!/bin/bash
# round 4 call 21: validation of the round's final build: full GPU suite, smoke(), the contract command (default: precision 5, PMC traffic, latency, cpu baseline),
# secondary lines, kernel trace of the contract workload
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1500 python bench.py > $O/bench_contract.json 2> $O/bench_contract.err; echo "contract rc=$?" >> $O/rc.txt
for c in qwen3 csm whisper; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
( cd tools; timeout 600 python bench_csm.py --no-cpu-baseline --weights fp8 > $O/bench_csm_fp8.json 2> $O/bench_csm_fp8.err; echo "csm fp8 rc=$?" >> $O/rc.txt
  timeout 600 python bench_qwen3.py --no-cpu-baseline --batch 1 --frames 32 > $O/bench_qwen3_b1.json 2> $O/bench_qwen3_b1.err; echo "qwen3 b1 rc=$?" >> $O/rc.txt )
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -8 $O/pytest_full.txt | cut -c1-200; tail -2 $O/smoke.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_contract.json")); r=d["roofline"]
print("contract", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "traffic ratio", r["traffic_over_algorithmic"] and round(r["traffic_over_algorithmic"],4), "lat", round(d["latency_b1"]["ms"],3), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"], "hbm frac", round(r["hbm_view"]["frac"],4))
for n in ("qwen3","csm","whisper","csm_fp8","qwen3_b1"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d.get("ms_per_step",0),2), "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "ttfb", d.get("ttfb_ms"))
    except Exception as e: print(n, "ERR", e)
PY
head -14 $O/kstats_b64.txt | cut -c1-150
