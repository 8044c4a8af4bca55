#!/bin/bash
# round-1 call 33: DAC / SNAC Snake prologues on conv_gemm's fast instantiation (in-kernel 1 / alpha): parity + the three codec lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_snac_gpu.py tests/test_dac_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/t_codec33.log 2>&1
echo "codec tests rc=$?"
timeout 300 python tools/bench_codecs.py > gpurun_out/bench_codecs_33.jsonl 2> gpurun_out/bench_codecs_33.err; echo "codecs rc=$?"
grep -E "exact_fp16_weights=True|passed|failed|Error" gpurun_out/t_codec33.log | cut -c1-260 | tail -8
python - <<'PY'
import json
for l in open("gpurun_out/bench_codecs_33.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["metric"][:60], "|", round(d["value"] / 1e6, 1), "M samples/s", round(d["x_realtime"]), "x RT", round(d["ms_per_step"], 2), "ms; conv", round(r["conv_gemm_ms"], 2), "ms", round(r["achieved"], 1), "TF/s")
PY
