#!/bin/bash
# round 6 call 8: (a) the full GPU suite on the build whose default mode is 6; (b) what a decode frame is made of: kernel traces of the Qwen3-TTS
# (64 and 8 utterances) and CSM frame loops, one frame each listed by tools/rocpd_timeline.py (launches, kernel time, gaps)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
for cfg in "qwen3 64" "qwen3 8" "csm 1"; do
  set -- $cfg; name=$1; b=$2
  timeout 400 rocprofv3 --kernel-trace -d $O/prof_${name}_$b -o t -- python $GRAFT_REPO_ROOT/tools/bench_$name.py --batch $b --frames 12 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/prof_${name}_$b.err; echo "trace $name $b rc=$?" >> $R
  DB=$(find $O/prof_${name}_$b -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then
    # one frame = the span between two talker / backbone first-codebook samples: marker = the sampling kernel; list enough occurrences back to skip the tail
    span=16; [ $name = csm ] && span=32
    python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py "$DB" sample_kernel $((span * 2 + 1)) --span=$span > $O/timeline_${name}_b$b.txt 2>&1
  fi
  rm -rf $O/prof_${name}_$b
done
cd $GRAFT_REPO_ROOT
cat $R; tail -4 $O/pytest_full.txt | cut -c1-250
for f in qwen3_b64 qwen3_b8 csm_b1; do echo "== $f"; head -30 $O/timeline_$f.txt | cut -c1-170; done
