#!/bin/bash
# round 6 call 25: one-pass k | v -> 16-bit head-major blocks for Whisper's cross-attention (mi355_kv_head_major16): parity, Whisper tests, the Whisper line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py tests/test_reference_fixtures_gpu.py -x -q > $O/pytest_whisper.txt 2>&1; echo "pytest rc=$?" >> $R
for i in 1 2; do
  timeout 400 python bench.py --config whisper 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['split_ms'])" >> $O/whisper_kvhm.txt
done
cat $R; tail -3 $O/pytest_whisper.txt | cut -c1-200; cat $O/whisper_kvhm.txt
