#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_new16.log 2>&1
echo "suite rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_v9.json 2> gpurun_out/bench_whisper_v9.err
echo "bench_whisper rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_w" -o w -- python "$GRAFT_REPO_ROOT/tools/bench_whisper.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_w.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_w.err"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_w -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 3 | cut -c1-200 > gpurun_out/whisper_kernel_stats_v9.txt 2>&1; rm -rf gpurun_out/prof_w
cat $R; tail -n 12 gpurun_out/t_new16.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_whisper_v9.json').read().strip().splitlines()[-1])
print('whisper value', round(d['value'],1), d['split_ms'], d['decode_ms_per_token_step'])
PY
head -n 9 gpurun_out/whisper_kernel_stats_v9.txt
