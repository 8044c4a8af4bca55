#!/bin/bash
# round 3 call 13: SNAC LocalMHA parity, full GPU suite after the fp8-tile / 5..8-row / call_struct changes, default bench (latency_b1 after the cheaper struct fill)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export MI355_MARGIN_REPORT=$O/margin_report.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/t_full.log 2>&1; echo "full rc=$?" > $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-pmc > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
tail -14 $O/t_full.log; cat $O/rc.txt; tail -1 $O/smoke.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_default.json")); print("kokoro", round(d["value"]/1e6,1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), d.get("latency_b1"))
PY
