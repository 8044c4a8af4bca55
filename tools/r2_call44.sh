#!/bin/bash
# round 2 call 44: final validation of the round + the bench lines and profiles that go into profiles/ (r2_*_call37)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export MI355_MARGIN_REPORT=$O/margin_report.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/t_full.log 2>&1; echo "full rc=$?" > $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 5 --warmup 2 --ragged --no-cpu-baseline --no-roofline > $O/bench_ragged.json 2> $O/bench_ragged.err; echo "ragged rc=$?" >> $O/rc.txt
for c in whisper qwen3 csm kitten; do
  timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline > $O/prof_k.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_k -name "*results.db" | head -1) 8 > $O/kstats_kokoro.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
tail -4 $O/t_full.log; tail -1 $O/smoke.log; cat $O/rc.txt; head -c 330 $O/bench_default.json; echo; head -12 $O/kstats_kokoro.txt
