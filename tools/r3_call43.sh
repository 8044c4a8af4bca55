#!/bin/bash
# round 3 calls 43-44: adain_from_partials with 64 block lanes per channel; 44: split rule from 24 steps for 129..255 tiles
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_edge_cases_gpu.py -q -m gpu > $O/t_k.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-roofline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline --no-latency > $O/prof_l.log 2>&1
  DB=$(find $O/prof_l -name "*results.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 23 --by-grid > $O/kstats_b1.txt; rm -rf $O/prof_l )
cat $O/rc.txt; tail -2 $O/t_k.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench.json")); print(round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "lat", round(d["latency_b1"]["ms"],3))
PY
head -3 $O/kstats_b1.txt | cut -c1-100; grep adain_from $O/kstats_b1.txt | cut -c1-130
