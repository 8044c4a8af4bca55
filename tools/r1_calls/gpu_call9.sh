#!/bin/bash
# round-1 call 9: native decode-step runner (Whisper decoder, Qwen3 talker / code predictor, CSM), conv_gemm EXT split (Kokoro regression fix)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py tests/test_transformer_kernels_gpu.py tests/test_mimi_gpu.py tests/test_qwen3_codec_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_new9.log 2>&1
echo "suite rc=$?" | tee -a $R
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_kokoro_v5.json 2> gpurun_out/bench_kokoro_v5.err
echo "bench kokoro rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_v4.json 2> gpurun_out/bench_whisper_v4.err
echo "bench_whisper rc=$?" | tee -a $R
timeout 600 python tools/bench_qwen3.py > gpurun_out/bench_qwen3_v3.json 2> gpurun_out/bench_qwen3_v3.err
echo "bench_qwen3 rc=$?" | tee -a $R
timeout 600 python tools/bench_csm.py > gpurun_out/bench_csm_v3.json 2> gpurun_out/bench_csm_v3.err
echo "bench_csm rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_q" -o q -- python "$GRAFT_REPO_ROOT/tools/bench_qwen3.py" --steps 1 --warmup 1 --frames 16 > "$GRAFT_REPO_ROOT/gpurun_out/prof_q.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_q.err"
echo "rocprof qwen3 rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_q -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 2 | cut -c1-200 > gpurun_out/qwen3_kernel_stats_v3.txt 2>&1; rm -rf gpurun_out/prof_q
cat $R; tail -n 30 gpurun_out/t_new9.log; cat gpurun_out/bench_kokoro_v5.json gpurun_out/bench_whisper_v4.json gpurun_out/bench_qwen3_v3.json gpurun_out/bench_csm_v3.json
for f in bench_kokoro_v5 bench_whisper_v4 bench_qwen3_v3 bench_csm_v3; do tail -n 3 gpurun_out/$f.err; done; head -n 16 gpurun_out/qwen3_kernel_stats_v3.txt
