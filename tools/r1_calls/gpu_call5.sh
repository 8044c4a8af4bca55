#!/bin/bash
# round-1 call 5: new transformer kernels + Whisper path (first GPU contact), regression of the Kokoro suite, whisper bench + profile
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "transformer_kernels rc=$?" | tee -a $R
timeout 900 python -m pytest tests/test_whisper_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_whisper.log 2>&1
echo "whisper rc=$?" | tee -a $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_old.log 2>&1
echo "old suite rc=$?" | tee -a $R
timeout 600 python tools/bench_whisper.py > gpurun_out/bench_whisper.json 2> gpurun_out/bench_whisper.err
echo "bench_whisper rc=$?" | tee -a $R
timeout 300 python tools/bench_whisper.py --batch 1 --no-cpu-baseline > gpurun_out/bench_whisper_b1.json 2>> gpurun_out/bench_whisper.err
echo "bench_whisper b1 rc=$?" | tee -a $R
timeout 300 python tools/bench_whisper.py --precision 3 --no-cpu-baseline > gpurun_out/bench_whisper_p3.json 2>> gpurun_out/bench_whisper.err
echo "bench_whisper p3 rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_whisper" -o w -- python "$GRAFT_REPO_ROOT/tools/bench_whisper.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_whisper.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_whisper.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_whisper -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 3 > gpurun_out/whisper_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_whisper
cat $R; tail -30 gpurun_out/t_kernels.log; tail -30 gpurun_out/t_whisper.log; tail -8 gpurun_out/t_old.log; cat gpurun_out/bench_whisper.json gpurun_out/bench_whisper_b1.json gpurun_out/bench_whisper_p3.json; tail -5 gpurun_out/bench_whisper.err; head -25 gpurun_out/whisper_kernel_stats.txt
