#!/bin/bash
# round 6 call 26: kernel-by-kernel statistics of the secondary lines (hunting launches that cost far more than their bytes / FLOPs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o s -- python "$@" > $O/bench_$name.json 2> $O/prof_$name.err; echo "$name rc=$?" >> $R
  DB=$(find $O/prof_$name -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$DB" 1 > $O/kernel_stats_$name.txt 2>/dev/null
  rm -rf $O/prof_$name
}
run qwen3_b64 $GRAFT_REPO_ROOT/tools/bench_qwen3.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline
run csm $GRAFT_REPO_ROOT/tools/bench_csm.py --steps 2 --warmup 1 --no-cpu-baseline
run kitten $GRAFT_REPO_ROOT/tools/bench_kitten.py --steps 3 --warmup 1 --no-cpu-baseline
run codecs $GRAFT_REPO_ROOT/tools/bench_codecs.py
cd $GRAFT_REPO_ROOT; cat $R
for n in qwen3_b64 csm kitten codecs; do echo "== $n"; head -22 $O/kernel_stats_$n.txt | cut -c1-170; tail -2 $O/prof_$n.err | cut -c1-200; done
