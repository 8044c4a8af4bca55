#!/bin/bash
# round 3 call 31: Whisper tall path with the fused self-attention step and planes out of the cross-attention: parity (whisper + every stack user), line at 64 / 32 windows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_whisper_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -q -m gpu > $O/t_whisper.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for b in 64 32; do
  timeout 900 python tools/bench_whisper.py --batch $b --no-cpu-baseline > $O/whisper_b$b.json 2> $O/whisper_b$b.err; echo "b$b rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; tail -6 $O/t_whisper.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("b32","b64"):
    try:
        d=json.load(open(O+"/whisper_%s.json"%n)); print(n, round(d["value"],1), "x RT  ms/step", round(d["ms_per_step"],2), d.get("split_ms"), "ms/token-step", round(d.get("decode_ms_per_token_step",0),3))
    except Exception as e: print(n, "ERR", e, open(O+"/whisper_%s.err"%n).read()[-500:])
PY
