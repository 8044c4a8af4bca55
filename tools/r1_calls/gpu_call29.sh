#!/bin/bash
# round-1 call 29 (final record): the complete GPU suite in one process (what the driver runs), smoke(), the contract bench line with roofline +
# cpu_baseline, rocprofv3 kernel stats of the same command, and the secondary decode lines (Whisper / Qwen3 / CSM bf16 + fp8)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full29.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke29.log 2>&1
echo "smoke rc=$?" | tee -a $R
timeout 600 python bench.py > gpurun_out/bench_final29.json 2> gpurun_out/bench_final29.err
echo "bench rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_k" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_k.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_k.err"
echo "rocprof kokoro rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_k -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 4 | cut -c1-200 > gpurun_out/kokoro_kernel_stats_29.txt 2>&1; rm -rf gpurun_out/prof_k
timeout 240 python tools/bench_whisper.py > gpurun_out/bench_whisper_29.json 2> gpurun_out/bench_whisper_29.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_29.json 2> gpurun_out/bench_qwen3_29.err; echo "qwen3 rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py > gpurun_out/bench_csm_29.json 2> gpurun_out/bench_csm_29.err; echo "csm rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py --weights fp8 > gpurun_out/bench_csm_fp8_29.json 2> gpurun_out/bench_csm_fp8_29.err; echo "csm fp8 rc=$?" | tee -a $R
cat $R; tail -n 12 gpurun_out/t_full29.log | cut -c1-250; tail -n 2 gpurun_out/smoke29.log; cat gpurun_out/bench_final29.json; tail -n 3 gpurun_out/bench_final29.err
head -n 10 gpurun_out/kokoro_kernel_stats_29.txt | cut -c1-160
python - <<'PY'
import json
for n in ("whisper_29", "qwen3_29", "csm_29", "csm_fp8_29"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, d.get("split_ms"), "frac", round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-300:])
PY
