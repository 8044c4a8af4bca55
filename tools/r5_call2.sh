#!/bin/bash
# round 5 call 2: (a) why do identical rows of a batch differ (call 1: 64 identical canonical utterances, mode 5: 4.9e-3 of the peak, spread 5e-2 over the rows)?
# identical-row diagnostic per intermediate tensor, modes 5 and 2, B = 64 / 4, run-to-run repeat; (b) the octet LSTM kernel: oracle tests, A/B timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python tools/diag_batch_rows.py --precision 5 --batch 64 --rows --repeat 2 > $O/diag_rows_p5_b64.txt 2>&1; echo "diag p5 b64 rc=$?" >> $R
timeout 300 python tools/diag_batch_rows.py --precision 2 --batch 64 --repeat 2 > $O/diag_rows_p2_b64.txt 2>&1; echo "diag p2 b64 rc=$?" >> $R
timeout 300 python tools/diag_batch_rows.py --precision 5 --batch 4 --rows --repeat 2 > $O/diag_rows_p5_b4.txt 2>&1; echo "diag p5 b4 rc=$?" >> $R
timeout 300 python tools/diag_batch_rows.py --precision 4 --batch 64 > $O/diag_rows_p4_b64.txt 2>&1; echo "diag p4 b64 rc=$?" >> $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm" > $O/pytest_lstm.txt 2>&1; echo "pytest lstm rc=$?" >> $R
timeout 300 python tools/bench_lstm.py > $O/bench_lstm.txt 2>&1; echo "bench lstm rc=$?" >> $R
cat $R; for f in diag_rows_p5_b64 diag_rows_p2_b64 diag_rows_p5_b4 diag_rows_p4_b64; do echo "== $f"; grep -v amdgpu.ids $O/$f.txt | cut -c1-400; done
tail -3 $O/pytest_lstm.txt | cut -c1-300; cat $O/bench_lstm.txt | grep -v amdgpu.ids
