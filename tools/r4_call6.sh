#!/bin/bash
# round 4 call 6: one-row GEMV: register-count-sized grids, DPP wave reductions, early norm-weight loads; parity of every GEMV user, CSM / Qwen3 / Whisper lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py -q -x > $O/pytest_lm.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
run() { name=$1; shift; timeout 600 env "$@" python bench_csm.py --no-cpu-baseline $NT > $O/csm_$name.json 2> $O/csm_$name.err; echo "$name rc=$?" >> $O/rc.txt; }
NT="" run base X=1
NT="--nt backbone,heads" run nt_bb X=1
NT="--nt backbone,heads --weights fp8" run nt_bb_fp8 X=1
cd ..
for c in qwen3 whisper; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt; done
cat $O/rc.txt; tail -3 $O/pytest_lm.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("base","nt_bb","nt_bb_fp8"):
    try:
        d=json.load(open(O+"/csm_%s.json"%n)); print(n, "ms/frame", round(d["ms_per_frame"],3), "x rt", round(d["value"],1), "frac", round(d["roofline"]["frac"],4))
    except Exception as e: print(n,"ERR",e, open(O+"/csm_%s.err"%n).read()[-300:])
for n in ("qwen3","whisper"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), {k:v for k,v in d.items() if k.startswith("ms_per")})
    except Exception as e: print(n, "ERR", e)
PY
