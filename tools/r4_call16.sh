#!/bin/bash
# round 4 call 16: one-wave decode attention for short key ranges + all prologue loads in flight, embed_sum, split groups of 2 steps: full GPU suite, secondary lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
timeout 600 python bench_csm.py --no-cpu-baseline > $O/csm_new.json 2> $O/csm_new.err; echo "csm rc=$?" >> $O/rc.txt
MI355_ATTN_ONE_WAVE=0 timeout 600 python bench_csm.py --no-cpu-baseline > $O/csm_4wave.json 2> $O/csm_4wave.err; echo "csm 4wave rc=$?" >> $O/rc.txt
timeout 600 python bench_csm.py --no-cpu-baseline --weights fp8 > $O/csm_fp8_new.json 2> $O/csm_fp8_new.err; echo "csm fp8 rc=$?" >> $O/rc.txt
cd ..
timeout 900 python bench.py --config qwen3 --no-cpu-baseline > $O/qwen3_b64_new.json 2> $O/qwen3_b64_new.err; echo "qwen3 b64 rc=$?" >> $O/rc.txt
MI355_ATTN_ONE_WAVE=0 timeout 900 python bench.py --config qwen3 --no-cpu-baseline > $O/qwen3_b64_4wave.json 2> $O/qwen3_b64_4wave.err; echo "qwen3 b64 4wave rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -8 $O/pytest_full.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("csm_new","csm_4wave","csm_fp8_new","qwen3_b64_new","qwen3_b64_4wave"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],2), d["unit"], "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "ttfb", d.get("ttfb_ms"))
    except Exception as e: print(n, "ERR", e)
PY
