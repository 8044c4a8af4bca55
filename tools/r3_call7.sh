#!/bin/bash
# round 3 call 7: (1) Qwen3 at 64 with the fused code-predictor SwiGLU; (2) VERDICT item 2 step (a): the Kokoro step at B = 64 under precision 1 / 2 / 3 / 4
# (how much of the conv time is MFMA issue), one workgroup per CU, and the per-shape table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/bench_qwen3.py --batch 64 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b64.json 2> $O/qwen3_b64.err; echo "q64 rc=$?" > $O/rc.txt
for p in 2 1 3 4; do
  timeout 600 python bench.py --steps 8 --warmup 3 --precision $p --no-pmc --no-cpu-baseline > $O/kokoro_p$p.json 2> $O/kokoro_p$p.err; echo "p$p rc=$?" >> $O/rc.txt
done
MI355_CONV_WS_WG_PER_CU=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/kokoro_wg1.json 2> $O/kokoro_wg1.err; echo "wg1 rc=$?" >> $O/rc.txt
MI355_CONV_WS_WG_PER_CU=3 timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/kokoro_wg3.json 2> $O/kokoro_wg3.err; echo "wg3 rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/qwen3_b64.json")); print('qwen3 b64', round(d['value'],1), d['split_ms'], round(d['ms_per_frame'],3), round(d['roofline']['frac'],4))
for n in ("p2","p1","p3","p4","wg1","wg3"):
    try:
        d=json.load(open(O+"/kokoro_%s.json"%n)); r=d.get("roofline") or {}
        print(n, "ms/step", round(d["ms_per_step"],2), "value", round(d["value"]/1e6,1), "conv_ms", r.get("conv_ms_per_step"), "frac", r.get("frac"), {k:r[k] for k in r if "ms" in k})
    except Exception as e: print(n, "ERR", e, open(O+"/kokoro_%s.err"%n).read()[-300:])
PY
