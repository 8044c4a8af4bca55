#!/bin/bash
# round 3 call 40: the LSTM recurrent product with its h reads requested in blocks (LSTM_BLK = 2: 16 exposed LDS round trips per step, 1: 32; before: 64):
# bit-exactness test, kernel time of the 64-utterance step and of one utterance
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
L=mlx_audio_amd/lib
for blk in 2 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLSTM_BLK=$blk -x hip -c mlx_audio_amd/csrc/lstm.hip -o $L/obj/lstm.o 2> $O/cc_$blk.err
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmi355audio.so $L/obj/*.o 2>> $O/cc_$blk.err; echo "build blk$blk rc=$?" >> $O/rc.txt
  timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "lstm" > $O/t_lstm_$blk.log 2>&1; echo "lstm blk$blk rc=$?" >> $O/rc.txt
  timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-roofline > $O/bench_blk$blk.json 2> $O/bench_blk$blk.err; echo "bench blk$blk rc=$?" >> $O/rc.txt
  ( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-latency --no-roofline > $O/prof_l.log 2>&1
    DB=$(find $O/prof_l -name "*results.db" | head -1); echo "BLK $blk: $(python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 | grep lstm | cut -c1-120)" >> $O/lstm_blk.txt; rm -rf $O/prof_l )
done
cat $O/rc.txt; cat $O/lstm_blk.txt; tail -2 $O/t_lstm_2.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("blk2","blk1"):
    d=json.load(open(O+"/bench_%s.json"%v)); print(v, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "lat", round(d["latency_b1"]["ms"],3))
PY
