#!/bin/bash
# round 4 call 5: one-row GEMV in ONE resident round (occupancy-sized grid, half-K instantiation), vectorised attention prologue, stream policy; per-kernel trace of the best setting
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py -q -x -k "attention_prologue or csm or gemv" > $O/pytest_lm.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
run() { name=$1; shift; timeout 600 env "$@" python bench_csm.py --no-cpu-baseline $NT > $O/csm_$name.json 2> $O/csm_$name.err; echo "$name rc=$?" >> $O/rc.txt; }
NT="" run base MI355_ATTN_IN_OPROJ=0
NT="" run fused MI355_ATTN_IN_OPROJ=1
NT="--nt backbone,heads" run base_nt_bb MI355_ATTN_IN_OPROJ=0
NT="--nt backbone,heads" run fused_nt_bb MI355_ATTN_IN_OPROJ=1
cd /tmp; export TMPDIR=/tmp
MI355_ATTN_IN_OPROJ=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_csm -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --nt backbone,heads --steps 1 --warmup 1 > $O/prof_csm.log 2>&1
DB=$(find $O/prof_csm -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_csm_bygrid.txt 2>&1 || python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 > $O/kstats_csm_bygrid.txt 2>&1
rm -rf $O/prof_csm
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -3 $O/pytest_lm.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("base","fused","base_nt_bb","fused_nt_bb"):
    try:
        d=json.load(open(O+"/csm_%s.json"%n)); print(n, "ms/frame", round(d["ms_per_frame"],3), "x rt", round(d["value"],1), "frac", round(d["roofline"]["frac"],4))
    except Exception as e: print(n,"ERR",e, open(O+"/csm_%s.err"%n).read()[-300:])
PY
head -16 $O/kstats_csm_bygrid.txt | cut -c1-170
