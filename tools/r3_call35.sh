#!/bin/bash
# round 3 call 35: interior polyphase epilogue (UP instantiations): parity (conv tests, every decoder with transposed convs, Kokoro), default line with / without, codec lines with / without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_mimi_gpu.py tests/test_encodec_gpu.py tests/test_reference_fixtures_gpu.py tests/test_kitten_gpu.py -q -m gpu > $O/t.log 2>&1; echo "tests rc=$?" > $O/rc.txt
ls tests | grep -E "dac|snac|vocos|bigvgan|qwen3_codec|codec" > $O/codec_tests.txt
timeout 1200 python -m pytest $(ls tests/test_*dac*gpu*.py tests/test_*snac*gpu*.py tests/test_*bigvgan*gpu*.py tests/test_*codec*gpu*.py 2>/dev/null) -q -m gpu > $O/t2.log 2>&1; echo "codec tests rc=$?" >> $O/rc.txt
for v in up noup; do
  if [ $v = noup ]; then export MI355_CONV_NO_UPFAST=1; else unset MI355_CONV_NO_UPFAST; fi
  timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-latency > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?" >> $O/rc.txt
  timeout 600 python tools/bench_codecs.py > $O/codecs_$v.jsonl 2> $O/codecs_$v.err; echo "codecs $v rc=$?" >> $O/rc.txt
done
unset MI355_CONV_NO_UPFAST
cat $O/rc.txt; tail -3 $O/t.log; tail -3 $O/t2.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("up","noup"):
    d=json.load(open(O+"/bench_%s.json"%v)); print(v, round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4))
    for l in open(O+"/codecs_%s.jsonl"%v):
        d=json.loads(l); print("  ", v, d.get("config",{}).get("workload","?")[:50], round(d["value"]/1e6,2), "M samples/s", round(d["ms_per_step"],3), "ms")
PY
