#!/bin/bash
# round 4 call 4: CSM-1B frame: attention as the o-proj GEMV's prologue (depth decoder) and the weight-stream cache policy (nt for backbone / heads / decoder)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py -q -x -k "attention_prologue or csm" > $O/pytest_lm.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
run() { name=$1; shift; timeout 600 env "$@" python bench_csm.py --no-cpu-baseline $NT > $O/csm_$name.json 2> $O/csm_$name.err; echo "$name rc=$?" >> $O/rc.txt; }
NT="" run base MI355_ATTN_IN_OPROJ=0
NT="" run fused MI355_ATTN_IN_OPROJ=1
NT="--nt backbone,heads" run fused_nt_bb MI355_ATTN_IN_OPROJ=1
NT="--nt backbone,heads,decoder" run fused_nt_all MI355_ATTN_IN_OPROJ=1
NT="--nt decoder" run fused_nt_dec MI355_ATTN_IN_OPROJ=1
cd ..
cat $O/rc.txt; tail -3 $O/pytest_lm.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("base","fused","fused_nt_bb","fused_nt_all","fused_nt_dec"):
    try:
        d=json.load(open(O+"/csm_%s.json"%n)); print(n, "ms/frame", round(d["ms_per_frame"],3), "x rt", round(d["value"],1), "frac", round(d["roofline"]["frac"],4))
    except Exception as e: print(n,"ERR",e, open(O+"/csm_%s.err"%n).read()[-300:])
PY
