#!/bin/bash
# round-1 call 6: decoder-stack glue kernels + sampler, generic TransformerStack (4 variants), Qwen3 codec decoder, Mimi decoder,
# Qwen3 talker frame loop, CSM generate_frame -- first GPU contact for all of them
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
for t in test_lm_kernels_gpu test_qwen3_codec_gpu test_mimi_gpu test_codec_lm_gpu; do
  timeout 900 python -m pytest tests/$t.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/$t.log 2>&1
  echo "$t rc=$?" | tee -a $R
done
cat $R
for t in test_lm_kernels_gpu test_qwen3_codec_gpu test_mimi_gpu test_codec_lm_gpu; do echo "=== $t"; tail -60 gpurun_out/$t.log; done
