#!/bin/bash
# round 6 call 10: the launch-by-launch list of one Qwen3-TTS frame at 64 utterances (which launches are row epilogues, and what stands in front of them)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_q -o t -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --batch 64 --frames 12 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/prof_q.err
DB=$(find $O/prof_q -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py "$DB" sample_kernel 40 --span=16 --from-start=272 --list > $O/timeline_qwen3_b64_list.txt 2>&1
rm -rf $O/prof_q
grep -n "+" $O/timeline_qwen3_b64_list.txt | sed -n 1,140p | cut -c1-150
