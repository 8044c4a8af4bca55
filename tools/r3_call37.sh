#!/bin/bash
# round 3 calls 37-38: LSTM recurrent weights as an exactly scaled IEEE-half image (v_fma_mix_f32); 38: + the gate pre-activation requested one step ahead
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "lstm" > $O/t_lstm.log 2>&1; echo "lstm rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_api_gpu.py tests/test_reference_fixtures_gpu.py -q -m gpu > $O/t_k.log 2>&1; echo "kokoro rc=$?" >> $O/rc.txt
for v in f16; do
  if [ $v = bf16 ]; then export MI355_LSTM_BF16=1; else unset MI355_LSTM_BF16; fi
  timeout 900 python bench.py --no-cpu-baseline --no-pmc > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?" >> $O/rc.txt
done
unset MI355_LSTM_BF16
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-pmc --no-cpu-baseline --no-latency > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 6 > $O/kstats_b64_lstm.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -3 $O/t_lstm.log; tail -3 $O/t_k.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("f16",):
    d=json.load(open(O+"/bench_%s.json"%v)); print(v, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "lat", round(d["latency_b1"]["ms"],3))
PY
grep lstm $O/kstats_b64_lstm.txt | cut -c1-140
