#!/bin/bash
# round 3 call 12: fp8 tile images in the rows pipeline (config[4]: fp8 GEMMs): parity, CSM at 8 sequences fp8 vs bf16, at 1 sequence; MI355_ROWS_MIN 1 / 2 probes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -q -m gpu -k "rows or fp8 or real_widths or csm or qwen3" > $O/t_fp8.log 2>&1; echo "fp8 rc=$?" > $O/rc.txt
for wt in bf16 fp8; do
  timeout 600 python tools/bench_csm.py --batch 8 --frames 32 --steps 2 --weights $wt --no-cpu-baseline > $O/csm_b8_$wt.json 2> $O/c_$wt.err; echo "csm8 $wt rc=$?" >> $O/rc.txt
  timeout 600 python tools/bench_csm.py --batch 1 --frames 32 --steps 2 --weights $wt --no-cpu-baseline > $O/csm_b1_$wt.json 2>> $O/c_$wt.err; echo "csm1 $wt rc=$?" >> $O/rc.txt
done
for m in 1 2; do
  MI355_ROWS_MIN=$m timeout 600 python tools/bench_csm.py --batch 1 --frames 32 --steps 2 --weights fp8 --no-cpu-baseline > $O/csm_b1_fp8_min$m.json 2> $O/c_min.err; echo "csm1 fp8 min$m rc=$?" >> $O/rc.txt
done
MI355_ROWS_MIN=1 timeout 600 python tools/bench_csm.py --batch 1 --frames 32 --steps 2 --weights bf16 --no-cpu-baseline > $O/csm_b1_bf16_min1.json 2>> $O/c_min.err
MI355_ROWS_MIN=2 timeout 600 python tools/bench_qwen3.py --batch 4 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b4_min2.json 2>> $O/c_min.err
timeout 600 python tools/bench_qwen3.py --batch 4 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b4_min5.json 2>> $O/c_min.err
tail -6 $O/t_fp8.log; cat $O/rc.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("csm_b8_bf16","csm_b8_fp8","csm_b1_bf16","csm_b1_fp8","csm_b1_fp8_min1","csm_b1_fp8_min2","csm_b1_bf16_min1","qwen3_b4_min2","qwen3_b4_min5"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_frame"],3))
    except Exception as e: print(n,"ERR",e)
PY
tail -n 5 $O/c_fp8.err
