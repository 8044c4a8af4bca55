#!/bin/bash
# round 6 call 14: GEMM-mode producers on the unmasked interior path: parity, the wide-GEMM A/B, Whisper bench (split on / off)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 600 python -m pytest tests/test_conv_split_gpu.py tests/test_kernels_gpu.py -x -q > $O/pytest_conv_split.txt 2>&1; echo "pytest conv_split+kernels rc=$?" >> $R
timeout 400 python tools/bench_conv.py --big-gemm --batch 64 --rounds 5 --out $O/conv_big_gemm_fastw_b64.txt > /dev/null 2> $O/conv_big_gemm.err; echo "big-gemm rc=$?" >> $R
timeout 900 python -m pytest tests/test_whisper_gpu.py -x -q > $O/pytest_whisper.txt 2>&1; echo "pytest whisper rc=$?" >> $R
timeout 600 python bench.py --config whisper > $O/bench_whisper_split.json 2> $O/bench_whisper.err; echo "bench whisper rc=$?" >> $R
MI355_WHISPER_SPLIT=0 timeout 600 python bench.py --config whisper > $O/bench_whisper_nosplit.json 2>> $O/bench_whisper.err; echo "bench whisper nosplit rc=$?" >> $R
cat $R; tail -5 $O/pytest_conv_split.txt | cut -c1-220; cat $O/conv_big_gemm_fastw_b64.txt; tail -4 $O/pytest_whisper.txt | cut -c1-200
python - <<'PY'
import json
for f in ("bench_whisper_split", "bench_whisper_nosplit"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["split_ms"], d["phase_rooflines"]["encoder"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
