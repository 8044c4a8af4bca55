#!/bin/bash
# round 4 call 20: gemv1_stream_kernel with the loop's last pair consumed without re-requesting a group and the requests pinned behind their group
# (sched_barrier): GEMV parity, CSM / Qwen3 one-sequence lines against the previous build on the same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; L=mlx_audio_amd/lib
timeout 1200 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py tests/test_codec_lm_gpu.py -q -x > $O/pytest_c20.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
for v in new prev new2; do
  if [ $v = prev ]; then cp ../$L/libmi355audio.so ../$L/libmi355audio_new.so; cp ../$L/libmi355audio_prev.so ../$L/libmi355audio.so; fi
  if [ $v = new2 ]; then cp ../$L/libmi355audio_new.so ../$L/libmi355audio.so; fi
  timeout 600 python bench_csm.py --no-cpu-baseline > $O/csm_$v.json 2> $O/csm_$v.err; echo "csm $v rc=$?" >> $O/rc.txt
  timeout 600 python bench_qwen3.py --no-cpu-baseline --batch 1 --frames 32 > $O/qwen3_b1_$v.json 2> $O/qwen3_b1_$v.err; echo "qwen3 b1 $v rc=$?" >> $O/rc.txt
done
cd ..
cat $O/rc.txt; tail -3 $O/pytest_c20.txt | cut -c1-200
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("csm_new","csm_prev","csm_new2","qwen3_b1_new","qwen3_b1_prev","qwen3_b1_new2"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],2), d["unit"], "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4))
    except Exception as e: print(n, "ERR", e)
PY
