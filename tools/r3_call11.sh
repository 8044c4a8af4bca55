#!/bin/bash
# round 3 call 11: rows pipeline for 5..8 sequences (16-row planes) against the matrix-pipe GEMV it replaces: Qwen3 at 8 utterances, CSM at 8 sequences, parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_whisper_gpu.py -q -m gpu > $O/t_lm.log 2>&1; echo "lm rc=$?" > $O/rc.txt
for m in 5 9; do
  MI355_ROWS_MIN=$m timeout 600 python tools/bench_qwen3.py --batch 8 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b8_min$m.json 2> $O/q.err; echo "q8 min$m rc=$?" >> $O/rc.txt
  MI355_ROWS_MIN=$m timeout 600 python tools/bench_csm.py --batch 8 --frames 32 --steps 1 --no-cpu-baseline > $O/csm_b8_min$m.json 2> $O/c.err; echo "csm8 min$m rc=$?" >> $O/rc.txt
done
tail -6 $O/t_lm.log; cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("qwen3_b8_min5","qwen3_b8_min9","csm_b8_min5","csm_b8_min9"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_frame"],3))
    except Exception as e: print(n,"ERR",e)
PY
tail -5 $O/q.err $O/c.err
