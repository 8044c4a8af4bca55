#!/bin/bash
# round 3 calls 15-16: split-K for launches of few tiles (one utterance per call): parity of the conv tests, the Kokoro tests, latency_b1 with / without (16: float4 finish)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv" > $O/t_conv.log 2>&1; echo "conv rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_kokoro_gpu.py tests/test_api_gpu.py -q -m gpu > $O/t_kokoro.log 2>&1; echo "kokoro rc=$?" >> $O/rc.txt
for v in 0 1; do
  MI355_CONV_SPLIT=$v timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/b1_split$v.json 2> $O/b1_split$v.err; echo "b1 split=$v rc=$?" >> $O/rc.txt
done
MI355_CONV_SPLIT_MINSTEPS=2 timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/b1_min2.json 2> $O/b1_min2.err; echo "b1 min2 rc=$?" >> $O/rc.txt
MI355_CONV_SPLIT_MINSTEPS=8 timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/b1_min8.json 2> $O/b1_min8.err; echo "b1 min8 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/prof_b1.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 41 --by-grid > $O/kstats_b1_split_bygrid.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -4 $O/t_conv.log; tail -4 $O/t_kokoro.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("split0","split1","min2","min8"):
    try:
        d=json.load(open(O+"/b1_%s.json"%n)); print(n, "ms/step", round(d["ms_per_step"],3), "latency_b1", d.get("latency_b1",{}).get("ms"))
    except Exception as e: print(n, "ERR", e, open(O+"/b1_%s.err"%n).read()[-400:])
PY
head -30 $O/kstats_b1_split_bygrid.txt | cut -c1-170
