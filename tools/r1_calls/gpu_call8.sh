#!/bin/bash
# round-1 call 8: GEMV prefetch-ring rewrite, fused q+k rope launch, two-step code-predictor prologue, PL-BERT on the flash kernel,
# left-padded attention; small-shape tile A/B; all four bench lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py tests/test_mimi_gpu.py tests/test_qwen3_codec_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_new8.log 2>&1
echo "suite rc=$?" | tee -a $R
timeout 300 python tools/bench_conv.py --small --out gpurun_out/conv_small_ab.txt > gpurun_out/conv_small.log 2>&1
echo "conv small rc=$?" | tee -a $R
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_kokoro_v4.json 2> gpurun_out/bench_kokoro_v4.err
echo "bench kokoro rc=$?" | tee -a $R
timeout 400 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_v3.json 2> gpurun_out/bench_whisper_v3.err
echo "bench_whisper rc=$?" | tee -a $R
timeout 600 python tools/bench_qwen3.py > gpurun_out/bench_qwen3_v2.json 2> gpurun_out/bench_qwen3_v2.err
echo "bench_qwen3 rc=$?" | tee -a $R
timeout 600 python tools/bench_csm.py > gpurun_out/bench_csm_v2.json 2> gpurun_out/bench_csm_v2.err
echo "bench_csm rc=$?" | tee -a $R
cat $R; tail -n 30 gpurun_out/t_new8.log; cat gpurun_out/conv_small_ab.txt; cat gpurun_out/bench_kokoro_v4.json gpurun_out/bench_whisper_v3.json gpurun_out/bench_qwen3_v2.json gpurun_out/bench_csm_v2.json
for f in bench_kokoro_v4 bench_whisper_v3 bench_qwen3_v2 bench_csm_v2; do tail -n 4 gpurun_out/$f.err; done
