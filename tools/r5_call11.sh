#!/bin/bash
# round 5 call 11: secondary lines on the round's build (per-kernel resident grids in gemv, -fno-slp-vectorize everywhere) + kernel trace of the contract workload
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
for c in qwen3 csm whisper kitten; do
  timeout 240 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $R
done
( cd tools; timeout 200 python bench_csm.py --no-cpu-baseline --weights fp8 > $O/bench_csm_fp8.json 2> $O/bench_csm_fp8.err; echo "csm fp8 rc=$?" >> $R
  timeout 200 python bench_qwen3.py --no-cpu-baseline --batch 1 --frames 32 > $O/bench_qwen3_b1.json 2> $O/bench_qwen3_b1.err; echo "qwen3 b1 rc=$?" >> $R )
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-latency --no-secondary-precision > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_d -o p -- python $GRAFT_REPO_ROOT/tools/bench_dsp.py --steps 10 --no-cpu-baseline > $O/prof_d.log 2>&1
DB=$(find $O/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 14 > $O/kstats_dsp.txt 2>&1
rm -rf $O/prof_d
cd $GRAFT_REPO_ROOT
cat $R
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("qwen3","csm","whisper","kitten","csm_fp8","qwen3_b1"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d.get("ms_per_step",0),2), "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "ttfb", d.get("ttfb_ms"))
    except Exception as e: print(n, "ERR", e)
PY
head -12 $O/kstats_b64.txt | cut -c1-160; head -6 $O/kstats_dsp.txt | cut -c1-160
