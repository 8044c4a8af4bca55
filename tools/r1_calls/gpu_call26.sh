#!/bin/bash
# round-1 call 26: matrix-pipe GEMV for 5..8 rows (gemv_mfma.hip): parity, then the complete suite, launch periods A/B (MI355_GEMV_MFMA=0), decode lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 300 python -m pytest tests/test_transformer_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemv" > gpurun_out/t_gemv26.log 2>&1
echo "gemv tests rc=$?" | tee -a $R
timeout 200 python tools/bench_gemv.py --tag mfma --iters 200 > gpurun_out/gemv26_mfma.txt 2>&1; echo "rc=$?" | tee -a $R
MI355_GEMV_MFMA=0 timeout 200 python tools/bench_gemv.py --tag fma --iters 200 > gpurun_out/gemv26_fma.txt 2>&1; echo "rc=$?" | tee -a $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full26.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_26.json 2> gpurun_out/bench_whisper_26.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_26.json 2> gpurun_out/bench_qwen3_26.err; echo "qwen3 rc=$?" | tee -a $R
cat $R; tail -n 25 gpurun_out/t_gemv26.log | cut -c1-250; tail -n 25 gpurun_out/t_full26.log | cut -c1-250
paste -d'\n' <(grep "us " gpurun_out/gemv26_mfma.txt | grep -v "^{") <(grep "us " gpurun_out/gemv26_fma.txt | grep -v "^{") | grep "M=8"
python - <<'PY'
import json
for n in ("whisper_26", "qwen3_26"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, d.get("split_ms"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-300:])
PY
