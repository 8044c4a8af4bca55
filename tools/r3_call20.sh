#!/bin/bash
# round 3 call 20: conv_ws4 on 128 x 64 tiles (thin outputs): parity, KittenTTS and codec lines with / without (MI355_CONV_NO_WS64=1), kernel trace of KittenTTS
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > $O/t_conv.log 2>&1; echo "conv rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_kitten_gpu.py -q -m gpu > $O/t_kitten.log 2>&1; echo "kitten rc=$?" >> $O/rc.txt
for v in ws64 nows64; do
  if [ $v = nows64 ]; then export MI355_CONV_NO_WS64=1; else unset MI355_CONV_NO_WS64; fi
  timeout 600 python tools/bench_kitten.py --no-cpu-baseline > $O/kitten_$v.json 2> $O/kitten_$v.err; echo "kitten $v rc=$?" >> $O/rc.txt
  timeout 600 python tools/bench_codecs.py > $O/codecs_$v.jsonl 2> $O/codecs_$v.err; echo "codecs $v rc=$?" >> $O/rc.txt
done
unset MI355_CONV_NO_WS64
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/tools/bench_kitten.py --no-cpu-baseline --steps 4 --warmup 2 > $O/prof_kitten.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 1 > $O/kstats_kitten.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -5 $O/t_conv.log; tail -5 $O/t_kitten.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("ws64","nows64"):
    try:
        d=json.load(open(O+"/kitten_%s.json"%v)); print(v, "kitten quant", round(d["value"]/1e6,1), "M  ms", round(d["ms_per_step"],2), " plain", round(d["without_activation_quant"]["value"]/1e6,1), "M ms", round(d["without_activation_quant"]["ms_per_step"],2), "conv_ms", d["roofline"].get("conv_ms_per_step"), "tflops", round(d["roofline"]["achieved"],1))
    except Exception as e: print(v, "ERR", e, open(O+"/kitten_%s.err"%v).read()[-400:])
    try:
        for l in open(O+"/codecs_%s.jsonl"%v):
            d=json.loads(l); print(v, d.get("config",{}).get("workload","?")[:50], round(d["value"]/1e6,2), "M samples/s", round(d["ms_per_step"],2), "ms")
    except Exception as e: print(v, "codecs ERR", e, open(O+"/codecs_%s.err"%v).read()[-400:])
PY
head -24 $O/kstats_kitten.txt | cut -c1-170
