#!/bin/bash
# round 4 call 14: generic conv kernel with its window rows / prologue operands in flight together, lean split-K finish: conv-side parity (Kokoro, KittenTTS,
# codecs, edge cases), Kokoro line + one-utterance latency with / without the lean finish, KittenTTS line, one-utterance kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_kokoro_gpu.py tests/test_kitten_gpu.py tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_conv_mx_gpu.py tests/test_vocos_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py tests/test_bigvgan_gpu.py tests/test_encodec_gpu.py tests/test_mimi_gpu.py tests/test_qwen3_codec_gpu.py tests/test_api_gpu.py tests/test_reference_fixtures_gpu.py -q -x > $O/pytest_c14.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $O/rc.txt
MI355_CONV_FINISH_OLD=1 timeout 900 python bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_p5_oldfinish.json 2> $O/bench_p5_oldfinish.err; echo "bench p5 oldfinish rc=$?" >> $O/rc.txt
timeout 900 python bench.py --config kitten --no-cpu-baseline > $O/bench_kitten.json 2> $O/bench_kitten.err; echo "kitten rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_l -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-latency --no-roofline > $O/prof_l.log 2>&1
DB=$(find $O/prof_l -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 23 --by-grid > $O/kstats_b1.txt 2>&1
rm -rf $O/prof_l
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -5 $O/pytest_c14.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("bench_p5","bench_p5_oldfinish"):
    try:
        d=json.load(open(O+"/%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "conv ms", round(r["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3), round(d["latency_b1"]["ms_min"],3))
    except Exception as e: print(n, "ERR", e)
try:
    d=json.load(open(O+"/bench_kitten.json")); print("kitten", round(d["value"]/1e6,1), "ms/step", round(d["ms_per_step"],3), d.get("quant_vs_plain"))
except Exception as e: print("kitten ERR", e)
PY
head -24 $O/kstats_b1.txt | cut -c1-170
