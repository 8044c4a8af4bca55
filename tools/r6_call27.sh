#!/bin/bash
# round 6 call 27: validation of the final tree -- full GPU suite, smoke(), the contract line with its in-run PMC passes, kernel trace of the same command, Whisper line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --no-roofline --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check > /dev/null 2> $O/prof_bench.err; echo "rocprof bench rc=$?" >> $R
DB=$(find $O/prof_bench -name "*_results.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 9 > $O/kernel_stats_b64.txt 2>/dev/null
rm -rf $O/prof_bench
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --config whisper > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "bench whisper rc=$?" >> $R
cat $R; tail -12 $O/pytest_full.txt | cut -c1-220; tail -2 $O/smoke.txt | cut -c1-250; cut -c1-2500 $O/bench_default.json; echo; head -12 $O/kernel_stats_b64.txt | cut -c1-170
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_whisper.json").read().strip().splitlines()[-1])
print("whisper", d["value"], d["ms_per_step"], d["split_ms"], d["roofline"]["frac"], d["phase_rooflines"]["encoder"]["frac"])
PY
