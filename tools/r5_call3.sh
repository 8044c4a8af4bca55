#!/bin/bash
# round 5 call 3: which launch first produces row-dependent outputs for identical utterances at B = 64 (call 2: F0 / N curves differ at 1e-2 in every mode)?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python tools/diag_batch_ops.py --precision 2 --batch 64 > $O/diag_ops_p2_b64.txt 2>&1; echo "diag ops p2 b64 rc=$?" >> $R
timeout 400 python tools/diag_batch_ops.py --precision 2 --batch 64 --poison --max-lines 12 > $O/diag_ops_p2_b64_poison.txt 2>&1; echo "diag ops p2 b64 poison rc=$?" >> $R
MI355_LSTM_OCT=0 timeout 400 python tools/diag_batch_ops.py --precision 2 --batch 64 --max-lines 8 > $O/diag_ops_p2_b64_oldlstm.txt 2>&1; echo "diag ops p2 b64 old lstm rc=$?" >> $R
timeout 400 python tools/diag_batch_ops.py --precision 2 --batch 32 --max-lines 8 > $O/diag_ops_p2_b32.txt 2>&1; echo "diag ops p2 b32 rc=$?" >> $R
timeout 400 python tools/diag_batch_ops.py --precision 5 --batch 4 --max-lines 12 > $O/diag_ops_p5_b4.txt 2>&1; echo "diag ops p5 b4 rc=$?" >> $R
cat $R; for f in diag_ops_p2_b64 diag_ops_p2_b64_poison diag_ops_p2_b64_oldlstm diag_ops_p2_b32 diag_ops_p5_b4; do echo "== $f"; grep -v amdgpu.ids $O/$f.txt | head -70 | cut -c1-330; done
