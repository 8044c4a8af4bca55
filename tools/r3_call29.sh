#!/bin/bash
# round 3 call 29: Whisper decoder on the rows pipeline (cross-attention in tall_step): parity, then the line at 8 / 8 (pipeline from 5) / 16 / 32 / 64 windows per step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_whisper_gpu.py -q -m gpu -x > $O/t_whisper.log 2>&1; echo "whisper rc=$?" > $O/rc.txt
for b in 8 16 32 64; do
  timeout 900 python tools/bench_whisper.py --batch $b --no-cpu-baseline > $O/whisper_b$b.json 2> $O/whisper_b$b.err; echo "b$b rc=$?" >> $O/rc.txt
done
MI355_WHISPER_ROWS_MIN=5 timeout 900 python tools/bench_whisper.py --batch 8 --no-cpu-baseline > $O/whisper_b8_min5.json 2> $O/whisper_b8_min5.err; echo "b8min5 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -8 $O/t_whisper.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("b8","b8_min5","b16","b32","b64"):
    try:
        d=json.load(open(O+"/whisper_%s.json"%n)); print(n, round(d["value"],1), "x RT  ms/step", round(d["ms_per_step"],2), d.get("split_ms"), "ms/token-step", round(d.get("decode_ms_per_token_step",0),3))
    except Exception as e: print(n, "ERR", e, open(O+"/whisper_%s.err"%n).read()[-500:])
PY
