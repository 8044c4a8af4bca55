#!/bin/bash
# round 6 call 16: the unmasked interior producer path on every non-quantising instantiation: conv parity, Kokoro parity, per-shape A/B at precision 2, contract line, kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_mx_gpu.py tests/test_conv_split_gpu.py tests/test_edge_cases_gpu.py tests/test_kokoro_gpu.py -x -q > $O/pytest_fastw.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 400 python tools/bench_conv.py --batch 64 --rounds 5 --out $O/conv_ab_fastw_b64.txt > /dev/null 2> $O/conv_ab.err; echo "conv ab rc=$?" >> $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --no-roofline --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check > /dev/null 2> $O/prof_bench.err; echo "rocprof bench rc=$?" >> $R
DB=$(find $O/prof_bench -name "*_results.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 9 > $O/kernel_stats_b64.txt 2>/dev/null
rm -rf $O/prof_bench
MI355_WHISPER_SPLIT=0 timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_w -o w -- python $GRAFT_REPO_ROOT/tools/bench_whisper.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper_prof.json 2> $O/prof_w.err; echo "whisper prof rc=$?" >> $R
DB=$(find $O/prof_w -name "*_results.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 4 > $O/kernel_stats_whisper.txt 2>/dev/null
rm -rf $O/prof_w
cd "$GRAFT_REPO_ROOT"
cat $R; tail -4 $O/pytest_fastw.txt | cut -c1-200; cat $O/conv_ab_fastw_b64.txt | grep " ws4 "; cut -c1-1500 $O/bench_default.json; echo; head -14 $O/kernel_stats_b64.txt | cut -c1-180; head -30 $O/kernel_stats_whisper.txt | cut -c1-180
