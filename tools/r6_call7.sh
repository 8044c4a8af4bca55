#!/bin/bash
# round 6 call 7: the build with precision 6 as the DEFAULT mode: full GPU suite, smoke(), the contract line with its in-run PMC passes, the kernel trace
# of the same command, ablations / timeline of the FP4 kernel, the secondary lines (whisper with whole-phase rooflines; qwen3 at 8 utterances)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
timeout 300 python tools/bench_conv.py --ablate --precision 6 --batch 64 --out $O/conv_ablate_p6_b64.txt > /dev/null 2> $O/conv_ablate.err; echo "ablate rc=$?" >> $R
timeout 300 python tools/conv_timeline.py --precision 6 --batch 64 --out $O/conv_timeline_p6_b64.txt > /dev/null 2> $O/conv_timeline.err; echo "timeline rc=$?" >> $R
timeout 400 python bench.py --config whisper --no-cpu-baseline > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "whisper rc=$?" >> $R
timeout 400 python tools/bench_qwen3.py --batch 8 --no-cpu-baseline > $O/bench_qwen3_b8.json 2> $O/bench_qwen3_b8.err; echo "qwen3 b8 rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --no-roofline --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check > /dev/null 2> $O/prof_bench.err; echo "rocprof bench rc=$?" >> $R
DB=$(find $O/prof_bench -name "*_results.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 9 > $O/kernel_stats_b64.txt 2>/dev/null
rm -rf $O/prof_bench
cd $GRAFT_REPO_ROOT
cat $R; tail -4 $O/pytest_full.txt | cut -c1-250; grep -E "margin rule" $O/pytest_full.txt | cut -c1-200; tail -1 $O/smoke.txt
cut -c1-3000 $O/bench_default.json; echo; head -12 $O/kernel_stats_b64.txt | cut -c1-180
cat $O/conv_ablate_p6_b64.txt | head -24; grep -E "^##|^8 tiles|producer|^    [1-2] " $O/conv_timeline_p6_b64.txt | cut -c1-330 | head -20
cut -c1-2500 $O/bench_whisper.json; echo; cut -c1-900 $O/bench_qwen3_b8.json
