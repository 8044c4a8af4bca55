#!/bin/bash
# round-1 call 27: gemv_mfma with the whole K slice in flight (ring 8) and K <= 2048 only: parity, launch periods, per-shape kernel durations of a Qwen3 / Whisper run
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 300 python -m pytest tests/test_transformer_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemv" > gpurun_out/t_gemv27.log 2>&1
echo "gemv tests rc=$?" | tee -a $R
timeout 200 python tools/bench_gemv.py --tag mfma8 --iters 200 > gpurun_out/gemv27_mfma.txt 2>&1; echo "rc=$?" | tee -a $R
timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_27.json 2> gpurun_out/bench_whisper_27.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_27.json 2> gpurun_out/bench_qwen3_27.err; echo "qwen3 rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_q" -o q -- python "$GRAFT_REPO_ROOT/tools/bench_qwen3.py" --steps 1 --warmup 1 --frames 16 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_q.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_q.err"
echo "rocprof qwen3 rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_w" -o w -- python "$GRAFT_REPO_ROOT/tools/bench_whisper.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_w.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_w.err"
echo "rocprof whisper rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_q -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 2 --by-grid | cut -c1-200 > gpurun_out/qwen3_kernel_stats_27.txt 2>&1; rm -rf gpurun_out/prof_q
DB=$(find gpurun_out/prof_w -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 3 --by-grid | cut -c1-200 > gpurun_out/whisper_kernel_stats_27.txt 2>&1; rm -rf gpurun_out/prof_w
cat $R; tail -n 5 gpurun_out/t_gemv27.log | cut -c1-250
grep "us " gpurun_out/gemv27_mfma.txt | grep -v "^{" | grep -E "logits|talker (qkv|gate)"
head -n 24 gpurun_out/qwen3_kernel_stats_27.txt | cut -c1-170; head -n 20 gpurun_out/whisper_kernel_stats_27.txt | cut -c1-170
python - <<'PY'
import json
for n in ("whisper_27", "qwen3_27"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, d.get("split_ms"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-300:])
PY
