#!/bin/bash
# round-1 call 23: fused-norm GEMV with ONE read of x (statistics from the staged LDS copy): parity tests, launch-period A/B per decode shape, Whisper / Qwen3 / CSM lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 400 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemv or stack" > gpurun_out/t_gemv23.log 2>&1
echo "gemv/stack tests rc=$?" | tee -a $R
timeout 200 python tools/bench_gemv.py --tag one_read > gpurun_out/gemv_one_read.txt 2>&1; echo "bench_gemv rc=$?" | tee -a $R
MI355_GEMV_TWO_READS=1 timeout 200 python tools/bench_gemv.py --tag two_reads > gpurun_out/gemv_two_reads.txt 2>&1; echo "bench_gemv(two) rc=$?" | tee -a $R
timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_23.json 2> gpurun_out/bench_whisper_23.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_23.json 2> gpurun_out/bench_qwen3_23.err; echo "qwen3 rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py > gpurun_out/bench_csm_23.json 2> gpurun_out/bench_csm_23.err; echo "csm rc=$?" | tee -a $R
cat $R; tail -n 8 gpurun_out/t_gemv23.log | cut -c1-250
paste -d'\n' <(grep -v "^{" gpurun_out/gemv_one_read.txt | grep "us ") <(grep -v "^{" gpurun_out/gemv_two_reads.txt | grep "us ")
python - <<'PY'
import json
for n in ("whisper", "qwen3", "csm"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}_23.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, d.get("split_ms"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}_23.err").read()[-300:])
PY
