#!/bin/bash
# round-1 call 7: fused-norm / split GEMV + 16-wave decode attention (re-test everything that uses them), Whisper bench after the fusion,
# first Qwen3-TTS-1.7B and CSM-1B bench lines + kernel stats
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py tests/test_mimi_gpu.py tests/test_qwen3_codec_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_new.log 2>&1
echo "new-suite rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_v2.json 2> gpurun_out/bench_whisper_v2.err
echo "bench_whisper rc=$?" | tee -a $R
timeout 600 python tools/bench_qwen3.py > gpurun_out/bench_qwen3.json 2> gpurun_out/bench_qwen3.err
echo "bench_qwen3 rc=$?" | tee -a $R
timeout 600 python tools/bench_csm.py > gpurun_out/bench_csm.json 2> gpurun_out/bench_csm.err
echo "bench_csm rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_q" -o q -- python "$GRAFT_REPO_ROOT/tools/bench_qwen3.py" --steps 1 --warmup 1 --frames 16 > "$GRAFT_REPO_ROOT/gpurun_out/prof_q.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_q.err"
echo "rocprof qwen3 rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_q -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 2 | cut -c1-200 > gpurun_out/qwen3_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_q
cat $R; tail -30 gpurun_out/t_new.log; cat gpurun_out/bench_whisper_v2.json gpurun_out/bench_qwen3.json gpurun_out/bench_csm.json; tail -5 gpurun_out/bench_whisper_v2.err gpurun_out/bench_qwen3.err gpurun_out/bench_csm.err; head -22 gpurun_out/qwen3_kernel_stats.txt
