#!/bin/bash
# round 5 call 18: evidence on the round's build -- kernel traces (by launch shape) of the decode lines, the ragged Kokoro line, Kokoro at 512 utterances per GPU
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 200 python bench.py --ragged --no-pmc --no-cpu-baseline --no-latency --no-secondary-precision --steps 10 > $O/bench_ragged.json 2> $O/bench_ragged.err; echo "bench ragged rc=$?" >> $R
timeout 300 python bench.py --batch 512 --no-pmc --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check --steps 5 --warmup 2 > $O/bench_b512.json 2> $O/bench_b512.err; echo "bench b512 rc=$?" >> $R
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_c.log 2>&1; echo "trace csm rc=$?" >> $R
DB=$(find $O/prof_c -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 --by-grid > $O/kstats_csm.txt 2>&1
rm -rf $O/prof_c
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --no-cpu-baseline --batch 64 --steps 1 --warmup 1 > $O/prof_q.log 2>&1; echo "trace qwen3 rc=$?" >> $R
DB=$(find $O/prof_q -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_qwen3_b64.txt 2>&1
rm -rf $O/prof_q
cd $GRAFT_REPO_ROOT
cat $R
python - <<'PY'
import json
for f in ("bench_ragged", "bench_b512"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"] / 1e6, 2), "M samples/s", round(d["ms_per_step"], 2), "ms/step", d["config"]["workload"][:60], "frac", round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "ERR", e)
PY
head -12 $O/kstats_csm.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
head -14 $O/kstats_qwen3_b64.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
