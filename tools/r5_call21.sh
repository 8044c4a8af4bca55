#!/bin/bash
# round 5 call 21: the one-launch LSTM step (gates in the step GEMM's epilogue, ABI 32): kernel test in both forms, every EnCodec test (decode, encode, Vocos
# features), then the EnCodec decode / encode lines with the fused step and with MI355_LSTM_SEQ_FUSED=0 (same box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 420 python -m pytest tests/test_encodec_gpu.py tests/test_codec_encode_gpu.py tests/test_reference_fixtures_gpu.py -q -m gpu -k "lstm or encodec or Encodec or vocos" > $O/pytest_lstm_fused.txt 2>&1; echo "pytest rc=$?" >> $R
for mode in 1 0; do
  MI355_LSTM_SEQ_FUSED=$mode timeout 200 python tools/bench_codecs.py --only encodec --batch 16 --seconds 10 --steps 5 --warmup 2 > $O/bench_encodec_decode_fused$mode.json 2> $O/bench_encodec_decode_fused$mode.err; echo "decode fused=$mode rc=$?" >> $R
  MI355_LSTM_SEQ_FUSED=$mode timeout 200 python tools/bench_codecs.py --encode --only encodec --batch 16 --seconds 10 --steps 5 --warmup 2 > $O/bench_encodec_encode_fused$mode.json 2> $O/bench_encodec_encode_fused$mode.err; echo "encode fused=$mode rc=$?" >> $R
done
cat $R
tail -3 $O/pytest_lstm_fused.txt | cut -c1-200
grep -E "^(FAILED|ERROR)|Error|assert " $O/pytest_lstm_fused.txt | head -12 | cut -c1-300
python - <<'PY'
import json
for f in ("decode_fused1", "decode_fused0", "encode_fused1", "encode_fused0"):
    try:
        d = json.loads(open(f"gpurun_out/bench_encodec_{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"] / 1e6, 1), "M samples/s", round(d["ms_per_step"], 2), "ms", round(d["x_realtime"]), "x rt")
    except Exception as e:
        print(f, "ERR", e)
PY
