#!/bin/bash
# round 4 call 3: precision 5 with the deeper operand pipeline (four hi fragment sets, two lo operand sets): parity, per-shape A/B, whole step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_conv_mx_gpu.py -q -x -s > $O/pytest_conv_mx.txt 2>&1; echo "conv_mx rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_kokoro_gpu.py -q -x -s -k "precision5" > $O/pytest_kokoro_p5.txt 2>&1; echo "kokoro_p5 rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_conv.py --prec-ab --batch 32 --out $O/conv_prec_ab_b32.txt > /dev/null 2> $O/conv_prec_ab.err; echo "prec_ab rc=$?" >> $O/rc.txt
for p in 5 2; do
  timeout 900 python bench.py --precision $p --no-pmc --no-cpu-baseline > $O/bench_p$p.json 2> $O/bench_p$p.err; echo "bench p$p rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; tail -4 $O/pytest_conv_mx.txt; grep "precision" $O/pytest_conv_mx.txt | head -12; grep "kokoro precision\|passed\|failed\|Error" $O/pytest_kokoro_p5.txt | head; grep "p5_\|p2_\|p3_" $O/conv_prec_ab_b32.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for p in (2,5):
    try:
        d=json.load(open(O+"/bench_p%d.json"%p)); r=d["roofline"]
        print("p%d"%p, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3))
    except Exception as e: print(p, "ERR", e)
PY
