#!/bin/bash
# round 5 call 19: the codec ENCODE sides on the GPU (DAC / SNAC / EnCodec / Vocos EncodecFeatures: tests/test_codec_encode_gpu.py + the four decode test
# files whose error tests changed), then the measured table VERDICT r4 item 7 asks for: CSM-1B at 8 and 64 sequences, bf16 vs fp8 weight images, and the
# fp8 matrix-pipe GEMV (gemv_mfma_fp8.hip) put back on the 5..8-row path (MI355_ROWS_MIN=9) against the rows pipeline that replaced it there
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 420 python -m pytest tests/test_codec_encode_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py tests/test_encodec_gpu.py tests/test_vocos_gpu.py -q -m gpu -s > $O/pytest_codec_encode.txt 2>&1; echo "pytest rc=$?" >> $R
for cfg in "8 bf16" "8 fp8" "64 bf16" "64 fp8"; do
  set -- $cfg
  timeout 150 python tools/bench_csm.py --batch $1 --weights $2 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b$1_$2.json 2> $O/bench_csm_b$1_$2.err; echo "csm b$1 $2 rc=$?" >> $R
done
MI355_ROWS_MIN=9 timeout 150 python tools/bench_csm.py --batch 8 --weights fp8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b8_fp8_mfma.json 2> $O/bench_csm_b8_fp8_mfma.err; echo "csm b8 fp8 (mfma gemv) rc=$?" >> $R
cat $R
grep -E "passed|failed|error" $O/pytest_codec_encode.txt | tail -3
grep -E "^(dac|snac|encodec) " $O/pytest_codec_encode.txt | cut -c1-230
grep -E "^(FAILED|ERROR)|Error|assert " $O/pytest_codec_encode.txt | head -20 | cut -c1-300
python - <<'PY'
import json
for f in ("b8_bf16", "b8_fp8", "b8_fp8_mfma", "b64_bf16", "b64_fp8"):
    try:
        d = json.load(open(f"gpurun_out/bench_csm_{f}.json")); print(f, round(d["ms_per_frame"], 3), "ms/frame", round(d["value"], 1), d["unit"], "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
