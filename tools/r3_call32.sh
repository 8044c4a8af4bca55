#!/bin/bash
# round 3 call 32: adain_from_partials without per-block float64 divisions: parity of its users, default line, Whisper default line (64 windows, cross-attention roofline leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_edge_cases_gpu.py -q -m gpu > $O/t_k.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" >> $O/rc.txt
timeout 900 python bench.py --config whisper --no-cpu-baseline > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "whisper rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/t_k.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_default.json")); print("default", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "roofline", round(d["roofline"]["frac"],4), "lat", d["latency_b1"]["ms"])
d=json.load(open(O+"/bench_whisper.json")); print("whisper", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), d["split_ms"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items()})
PY
