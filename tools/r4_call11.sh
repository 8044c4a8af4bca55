#!/bin/bash
# round 4 call 11: straight-line one-row GEMV kernels (gemv1_stream_kernel, second gemv1_splitk_kernel): parity, CSM / Qwen3 b1 A/B against the old
# kernels on the same box (MI355_GEMV1_OLD=1), CSM kernel trace; Kokoro one-utterance latency with / without the float4 noise-conv path
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_whisper_gpu.py -q -x > $O/pytest_c11.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
for v in new old; do
  E=0; [ $v = old ] && E=1
  MI355_GEMV1_OLD=$E timeout 600 python bench_csm.py --no-cpu-baseline > $O/csm_$v.json 2> $O/csm_$v.err; echo "csm $v rc=$?" >> $O/rc.txt
  MI355_GEMV1_OLD=$E timeout 600 python bench_csm.py --no-cpu-baseline --weights fp8 > $O/csm_fp8_$v.json 2> $O/csm_fp8_$v.err; echo "csm fp8 $v rc=$?" >> $O/rc.txt
  MI355_GEMV1_OLD=$E timeout 600 python bench_qwen3.py --no-cpu-baseline --batch 1 --frames 32 > $O/qwen3_b1_$v.json 2> $O/qwen3_b1_$v.err; echo "qwen3 b1 $v rc=$?" >> $O/rc.txt
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_c.log 2>&1
DB=$(find $O/prof_c -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 > $O/kstats_csm.txt 2>&1
rm -rf $O/prof_c
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  MI355_CONV_NOVEC_FLAT=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline > $O/kokoro_novec$v.json 2> $O/kokoro_novec$v.err; echo "kokoro novec=$v rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; tail -4 $O/pytest_c11.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("csm_new","csm_old","csm_fp8_new","csm_fp8_old","qwen3_b1_new","qwen3_b1_old"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],2), d["unit"], "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "ttfb", d.get("ttfb_ms"))
    except Exception as e: print(n, "ERR", e)
for v in (0,1):
    try:
        d=json.load(open(O+"/kokoro_novec%d.json"%v)); print("kokoro novec", v, round(d["ms_per_step"],3), "conv", round(d["roofline"]["conv_gemm_ms_per_step"],2), "lat", round(d["latency_b1"]["ms"],3))
    except Exception as e: print("kokoro", v, "ERR", e)
PY
head -24 $O/kstats_csm.txt | cut -c1-170
