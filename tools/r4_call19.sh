#!/bin/bash
# round 4 call 19: conv_ws4 producers with the prologue coefficients requested with their window and every pass's load unconditional (exact waits: two windows
# of prefetch): parity, then a same-box A/B against the previous build (lib/libmi355audio_prev.so) on the per-shape table and the Kokoro line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; L=mlx_audio_amd/lib
timeout 1200 python -m pytest tests/test_conv_mx_gpu.py tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_edge_cases_gpu.py -q -x > $O/pytest_c19.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/bench_conv.py --prec-ab --batch 32 --out $O/conv_prec_ab_new.txt > /dev/null 2> $O/conv_prec_ab_new.err; echo "prec_ab new rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; echo "bench new rc=$?" >> $O/rc.txt
cp $L/libmi355audio.so $L/libmi355audio_new.so; cp $L/libmi355audio_prev.so $L/libmi355audio.so
timeout 600 python tools/bench_conv.py --prec-ab --batch 32 --out $O/conv_prec_ab_prev.txt > /dev/null 2> $O/conv_prec_ab_prev.err; echo "prec_ab prev rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_prev.json 2> $O/bench_prev.err; echo "bench prev rc=$?" >> $O/rc.txt
cp $L/libmi355audio_new.so $L/libmi355audio.so
timeout 900 python bench.py --no-pmc --no-cpu-baseline --no-latency > $O/bench_new2.json 2> $O/bench_new2.err; echo "bench new2 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_c19.txt | cut -c1-200
for v in new prev; do echo "== $v"; grep "p5_\|p2_" $O/conv_prec_ab_$v.txt | awk '{print $1,$2,$3,$4,$7,$8,$11}'; done
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("bench_new","bench_prev","bench_new2"):
    try:
        d=json.load(open(O+"/%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", d.get("latency_b1") and round(d["latency_b1"]["ms"],3))
    except Exception as e: print(n, "ERR", e)
PY
