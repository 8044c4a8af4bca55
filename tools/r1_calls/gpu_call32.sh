#!/bin/bash
# round-1 call 32: dilated-comb depthwise kernel: parity of its users (SNAC, Qwen3 codec ConvNeXt, Mimi), SNAC line again
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_snac_gpu.py tests/test_qwen3_codec_gpu.py tests/test_mimi_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/t_dw32.log 2>&1
echo "dwconv users rc=$?"
timeout 200 python tools/bench_codecs.py --only snac > gpurun_out/bench_snac_32.jsonl 2> gpurun_out/bench_snac_32.err; echo "snac rc=$?"
grep -E "snac depthwise=True noise=True|passed|failed|Error" gpurun_out/t_dw32.log | cut -c1-330 | tail -8
python - <<'PY'
import json
for l in open("gpurun_out/bench_snac_32.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["metric"][:60], "|", round(d["value"] / 1e6, 1), "M samples/s", round(d["x_realtime"]), "x RT", round(d["ms_per_step"], 2), "ms; conv", round(r["conv_gemm_ms"], 2), "ms")
PY
