#!/bin/bash
# round 3 call 42: validation of the build after the LSTM change (scaled IEEE-half recurrent weights): complete GPU suite, smoke(), the contract command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1500 python bench.py > $O/bench_contract.json 2> $O/bench_contract.err; echo "contract rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_gpu_full.txt; tail -1 $O/smoke.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_contract.json")); r=d["roofline"]
print("contract", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "traffic ratio", round(r["traffic_over_algorithmic"],4), "lat", round(d["latency_b1"]["ms"],3), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
PY
