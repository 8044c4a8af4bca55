#!/bin/bash
# round 4 call 10: float4 noise convs (padded harmonic rows, flat + vec path), DPP reductions in rows_finish: Kokoro / KittenTTS / kernel parity, contract-like bench, Qwen3 line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -q -x > $O/pytest_b.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --config qwen3 --no-cpu-baseline > $O/bench_qwen3.json 2> $O/bench_qwen3.err; echo "qwen3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --config kitten --no-cpu-baseline > $O/bench_kitten.json 2> $O/bench_kitten.err; echo "kitten rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest_b.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_p5.json")); r=d["roofline"]
print("p5", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3))
for n in ("qwen3","kitten"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d.items() if k.startswith("ms_per_f")}, d.get("quant_vs_plain"))
    except Exception as e: print(n, "ERR", e)
PY
