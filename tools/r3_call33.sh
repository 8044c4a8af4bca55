#!/bin/bash
# round 3 call 33: CSM-1B at 8 / 16 / 32 / 64 sequences per step (bf16 and fp8 tile images at 64)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for b in 16 32 64; do
  timeout 900 python tools/bench_csm.py --batch $b > $O/csm_b$b.json 2> $O/csm_b$b.err; echo "b$b rc=$?" >> $O/rc.txt
done
timeout 900 python tools/bench_csm.py --batch 64 --weights fp8 > $O/csm_b64_fp8.json 2> $O/csm_b64_fp8.err; echo "b64 fp8 rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("b16","b32","b64","b64_fp8"):
    try:
        d=json.load(open(O+"/csm_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4))
    except Exception as e: print(n, "ERR", e, open(O+"/csm_%s.err"%n).read()[-600:])
PY
