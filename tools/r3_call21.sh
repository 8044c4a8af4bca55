#!/bin/bash
# round 3 call 21: quantiser folded into the conv prologue (pre_fq): parity, KittenTTS tests, bench with / without (MI355_FOLD_QUANT=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > $O/t_conv.log 2>&1; echo "conv rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_kitten_gpu.py -q -m gpu > $O/t_kitten.log 2>&1; echo "kitten rc=$?" >> $O/rc.txt
for v in fold nofold; do
  if [ $v = nofold ]; then export MI355_FOLD_QUANT=0; else unset MI355_FOLD_QUANT; fi
  timeout 600 python tools/bench_kitten.py --no-cpu-baseline > $O/kitten_$v.json 2> $O/kitten_$v.err; echo "kitten $v rc=$?" >> $O/rc.txt
done
unset MI355_FOLD_QUANT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/tools/bench_kitten.py --no-cpu-baseline --steps 4 --warmup 2 > $O/prof_kitten.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 1 > $O/kstats_kitten_fold.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -12 $O/t_conv.log; tail -5 $O/t_kitten.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("fold","nofold"):
    try:
        d=json.load(open(O+"/kitten_%s.json"%v)); print(v, "kitten quant", round(d["value"]/1e6,1), "M  ms", round(d["ms_per_step"],2), " plain", round(d["without_activation_quant"]["value"]/1e6,1), "M ms", round(d["without_activation_quant"]["ms_per_step"],2), "conv_ms", d["roofline"].get("conv_ms_per_step"), "tflops", round(d["roofline"]["achieved"],1))
    except Exception as e: print(v, "ERR", e, open(O+"/kitten_%s.err"%v).read()[-400:])
PY
head -24 $O/kstats_kitten_fold.txt | cut -c1-170
