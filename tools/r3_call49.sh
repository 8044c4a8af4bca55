#!/bin/bash
# round 3 call 49: smoke() on the rebuilt library (ABI 28) + the speaker encoder's time per reference clip
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/rc.txt
timeout 100 python tools/bench_speaker_encoder.py --out $O/speaker_encoder.jsonl > $O/spk.log 2>&1; echo "spk rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/smoke.log; tail -8 $O/spk.log
