#!/bin/bash
# round 4 call 23: fake_quant_extrema with four rows in flight per thread: KittenTTS / quantiser parity, KittenTTS line, kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kitten_gpu.py tests/test_kernels_gpu.py -q -k "kitten or quant or fq or extrema" > $O/pytest_c23.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python bench.py --config kitten --no-cpu-baseline > $O/bench_kitten.json 2> $O/bench_kitten.err; echo "kitten rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitten --no-cpu-baseline --steps 3 --warmup 1 > $O/prof_kt.log 2>&1
DB=$(find $O/prof_kt -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 1 > $O/kstats_kitten.txt 2>&1
rm -rf $O/prof_kt
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -3 $O/pytest_c23.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_kitten.json")); print("kitten", round(d["value"]/1e6,1), "ms/step", round(d["ms_per_step"],3), {k:v for k,v in d.items() if "plain" in k or "quant" in k})
PY
head -14 $O/kstats_kitten.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
