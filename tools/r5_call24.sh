#!/bin/bash
# round 5 call 24: kernel trace of the KittenTTS quantised step with the extrema partials (which sweeps are left, what the quantising convs cost now)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
cd /tmp; export TMPDIR=/tmp
for mode in 1 0; do
  MI355_EXT_PARTIALS=$mode timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_k$mode -o p -- python $GRAFT_REPO_ROOT/tools/bench_kitten.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_k$mode.log 2>&1; echo "trace kitten ext=$mode rc=$?" >> $R
  DB=$(find $O/prof_k$mode -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 1 > $O/kstats_kitten_ext$mode.txt 2>&1
  rm -rf $O/prof_k$mode
done
cd $GRAFT_REPO_ROOT
cat $R
for mode in 1 0; do echo "== ext=$mode"; head -16 $O/kstats_kitten_ext$mode.txt | sed 's/(anonymous namespace):://g; s/void //; s/mi355conv:://g' | cut -c1-150; done
