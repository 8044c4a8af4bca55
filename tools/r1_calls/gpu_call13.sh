#!/bin/bash
# round-1 call 13: latency fixes (decode attention: whole key row / 8 value rows in flight; GEMV: weights prefetched first, statistics from
# registers, batched staging loads)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_new13.log 2>&1
echo "suite rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_v6.json 2> gpurun_out/bench_whisper_v6.err
echo "bench_whisper rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline --batch 1 > gpurun_out/bench_whisper_v6_b1.json 2>> gpurun_out/bench_whisper_v6.err
echo "bench_whisper b1 rc=$?" | tee -a $R
timeout 600 python tools/bench_qwen3.py > gpurun_out/bench_qwen3_v5.json 2> gpurun_out/bench_qwen3_v5.err
echo "bench_qwen3 rc=$?" | tee -a $R
timeout 600 python tools/bench_csm.py > gpurun_out/bench_csm_v5.json 2> gpurun_out/bench_csm_v5.err
echo "bench_csm rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_w" -o w -- python "$GRAFT_REPO_ROOT/tools/bench_whisper.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_w.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_w.err"
echo "rocprof whisper rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_q" -o q -- python "$GRAFT_REPO_ROOT/tools/bench_qwen3.py" --steps 1 --warmup 1 --frames 16 > "$GRAFT_REPO_ROOT/gpurun_out/prof_q.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_q.err"
echo "rocprof qwen3 rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_w -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 3 | cut -c1-200 > gpurun_out/whisper_kernel_stats_v6.txt 2>&1; rm -rf gpurun_out/prof_w
DB=$(find gpurun_out/prof_q -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 2 | cut -c1-200 > gpurun_out/qwen3_kernel_stats_v5.txt 2>&1; rm -rf gpurun_out/prof_q
cat $R; tail -n 20 gpurun_out/t_new13.log
for f in bench_whisper_v6 bench_whisper_v6_b1 bench_qwen3_v5 bench_csm_v5; do python - "$f" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/'+sys.argv[1]+'.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],1), d.get('split_ms'), d.get('decode_ms_per_token_step', d.get('ms_per_frame')), 'roofline', round(d['roofline']['achieved'],1), d['roofline']['unit'], round(d['roofline']['frac'],3))
PY
done
head -n 10 gpurun_out/whisper_kernel_stats_v6.txt; head -n 8 gpurun_out/qwen3_kernel_stats_v5.txt
