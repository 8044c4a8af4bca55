#!/bin/bash
# round 6 call 28: LayerNorm with its per-channel operands requested up front (202 -> 130 us at 96 000 x 768): the engines that use it, the Whisper line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_kernels_gpu.py tests/test_whisper_gpu.py tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_frontends_gpu.py tests/test_vocos_gpu.py -x -q > $O/pytest_ln.txt 2>&1; echo "pytest rc=$?" >> $R
python tools/ln_bench.py > $O/ln_bench.txt 2>&1
for i in 1 2; do timeout 400 python bench.py --config whisper 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['split_ms'])" >> $O/whisper_ln.txt; done
cat $R; tail -3 $O/pytest_ln.txt | cut -c1-200; cat $O/ln_bench.txt | tail -6; cat $O/whisper_ln.txt
