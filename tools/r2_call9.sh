#!/bin/bash
# round 2 call 9: re-run fixed tests, bench with in-run PMC, ragged bench, secondary configs through bench.py, kernel traces of the decode loops
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export MI355_MARGIN_REPORT=$O/margin_report.txt
timeout 600 python -m pytest tests/test_interpolate_gpu.py tests/test_whisper_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py tests/test_shard_nccl_gpu.py -q -m gpu > $O/t_new.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_kokoro.json 2> $O/bench_kokoro.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 3 --warmup 1 --ragged --no-cpu-baseline --no-roofline > $O/bench_ragged.json 2> $O/bench_ragged.err; echo "ragged rc=$?" >> $O/rc.txt
for c in csm qwen3 whisper; do
  timeout 600 python bench.py --config $c --steps 2 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
cd /tmp; export TMPDIR=/tmp
for c in csm qwen3 whisper; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_$c.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_$c -name "*results.db" | head -1) > $O/kstats_$c.txt 2>&1
  rm -rf $O/prof_$c
done
cd $GRAFT_REPO_ROOT
tail -3 $O/t_new.log; cat $O/rc.txt; head -c 600 $O/bench_kokoro.json
