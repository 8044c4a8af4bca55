#!/bin/bash
# round-1 call 25: GEMV staging with one norm weight / bias load per column (fewer registers: 3-4 waves per SIMD at 8 rows): parity, launch periods (NC = default | 1), decode lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 400 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemv or stack" > gpurun_out/t_gemv25.log 2>&1
echo "gemv/stack tests rc=$?" | tee -a $R
timeout 200 python tools/bench_gemv.py --tag nc_default --iters 200 > gpurun_out/gemv25_default.txt 2>&1; echo "rc=$?" | tee -a $R
MI355_GEMV_NC=1 timeout 200 python tools/bench_gemv.py --tag nc1 --iters 200 > gpurun_out/gemv25_nc1.txt 2>&1; echo "rc=$?" | tee -a $R
timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_25.json 2> gpurun_out/bench_whisper_25.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_25.json 2> gpurun_out/bench_qwen3_25.err; echo "qwen3 rc=$?" | tee -a $R
MI355_GEMV_NC=1 timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_25_nc1.json 2> gpurun_out/bench_qwen3_25_nc1.err; echo "qwen3 nc1 rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py > gpurun_out/bench_csm_25.json 2> gpurun_out/bench_csm_25.err; echo "csm rc=$?" | tee -a $R
cat $R; tail -n 8 gpurun_out/t_gemv25.log | cut -c1-250
paste -d'\n' <(grep "us " gpurun_out/gemv25_default.txt | grep -v "^{") <(grep "us " gpurun_out/gemv25_nc1.txt | grep -v "^{") | grep -E "logits|talker|codepred (qkv|gate)|csm bb"
python - <<'PY'
import json
for n in ("whisper_25", "qwen3_25", "qwen3_25_nc1", "csm_25"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, d.get("split_ms"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-300:])
PY
