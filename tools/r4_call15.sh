#!/bin/bash
# round 4 call 15: LSTM recurrent product on v_dot2c_f32_f16 (h as two packed fp16 planes): Kokoro / KittenTTS / kernel parity, Kokoro line with the one-utterance
# latency, split-K group size A/B (MI355_CONV_SPLIT_MINSTEPS) on the latency leg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_api_gpu.py tests/test_reference_fixtures_gpu.py tests/test_encodec_gpu.py -q > $O/pytest_c15.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $O/rc.txt
for ms in 2 1; do
  MI355_CONV_SPLIT_MINSTEPS=$ms timeout 900 python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_p5_ms$ms.json 2> $O/bench_p5_ms$ms.err; echo "bench p5 minsteps $ms rc=$?" >> $O/rc.txt
done
timeout 900 python bench.py --config kitten --no-cpu-baseline > $O/bench_kitten.json 2> $O/bench_kitten.err; echo "kitten rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-latency --no-roofline > $O/prof_l.log 2>&1
DB=$(find $O/prof_l -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 23 --by-grid > $O/kstats_b1.txt 2>&1
rm -rf $O/prof_l
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -5 $O/pytest_c15.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("bench_p5","bench_p5_ms2","bench_p5_ms1"):
    try:
        d=json.load(open(O+"/%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3), round(d["latency_b1"]["ms_min"],3))
    except Exception as e: print(n, "ERR", e)
try:
    d=json.load(open(O+"/bench_kitten.json")); print("kitten", round(d["value"]/1e6,1), "ms/step", round(d["ms_per_step"],3))
except Exception as e: print("kitten ERR", e)
PY
head -12 $O/kstats_b1.txt | cut -c1-170
