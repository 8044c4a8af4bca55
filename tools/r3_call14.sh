#!/bin/bash
# round 3 call 14: (1) what the matrix pipe of this box sustains (random vs zero operands: power / clock evidence for the conv ceiling);
# (2) per-kernel time of ONE utterance per call (the latency_b1 leg): where the 12.7 ms go
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 tools/bin/mfma_peak 2 50 1000 > $O/mfma_peak_random.jsonl 2> $O/mfma_peak.err; echo "peak rc=$?" > $O/rc.txt
MI355_MFMA_ZERO=1 timeout 120 tools/bin/mfma_peak 2 50 1000 > $O/mfma_peak_zero.jsonl 2>> $O/mfma_peak.err; echo "peak0 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/prof_b1.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 41 > $O/kstats_b1.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 41 --by-grid > $O/kstats_b1_bygrid.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; cat $O/mfma_peak_random.jsonl $O/mfma_peak_zero.jsonl | cut -c1-260; tail -2 $O/prof_b1.log | cut -c1-600; head -40 $O/kstats_b1.txt | cut -c1-180
