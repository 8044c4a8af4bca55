#!/bin/bash
# round 3 call 41: 256-wide LSTM with two gate rows per thread (half the LDS reads of h) on the exactly scaled IEEE-half image: parity, kernel time, lines (vs MI355_LSTM_ONE_ROW=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "lstm" > $O/t_lstm.log 2>&1; echo "lstm rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_api_gpu.py tests/test_reference_fixtures_gpu.py tests/test_edge_cases_gpu.py -q -m gpu > $O/t_k.log 2>&1; echo "kokoro rc=$?" >> $O/rc.txt
for v in two one; do
  if [ $v = one ]; then export MI355_LSTM_ONE_ROW=1; else unset MI355_LSTM_ONE_ROW; fi
  timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?" >> $O/rc.txt
done
unset MI355_LSTM_ONE_ROW
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-latency --no-roofline > $O/prof_l.log 2>&1
  DB=$(find $O/prof_l -name "*results.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 | grep lstm | cut -c1-120 > $O/lstm_two.txt; rm -rf $O/prof_l )
cat $O/rc.txt; cat $O/lstm_two.txt; tail -2 $O/t_lstm.log; tail -2 $O/t_k.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("two","one"):
    d=json.load(open(O+"/bench_%s.json"%v)); print(v, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "lat", round(d["latency_b1"]["ms"],3))
PY
