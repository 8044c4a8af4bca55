#!/bin/bash
# round 4 call 12: straight-line row epilogue (rows_finish_lean_kernel), decode attention with its operands requested up front (norm weights / rotary
# entries with the raw q|k|v, the chunk's first value rows with its keys, key passes in groups): parity + same-box A/B of the secondary lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_whisper_gpu.py tests/test_kernels_gpu.py tests/test_transformer_kernels_gpu.py tests/test_tts_model_protocol_gpu.py tests/test_mimi_gpu.py tests/test_qwen3_codec_gpu.py -q -x > $O/pytest_c12.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
cd tools
timeout 600 python bench_csm.py --no-cpu-baseline > $O/csm_new.json 2> $O/csm_new.err; echo "csm rc=$?" >> $O/rc.txt
timeout 600 python bench_csm.py --no-cpu-baseline --weights fp8 > $O/csm_fp8_new.json 2> $O/csm_fp8_new.err; echo "csm fp8 rc=$?" >> $O/rc.txt
timeout 600 python bench_qwen3.py --no-cpu-baseline --batch 1 --frames 32 > $O/qwen3_b1_new.json 2> $O/qwen3_b1_new.err; echo "qwen3 b1 rc=$?" >> $O/rc.txt
cd ..
timeout 900 python bench.py --config qwen3 --no-cpu-baseline > $O/qwen3_b64_new.json 2> $O/qwen3_b64_new.err; echo "qwen3 b64 rc=$?" >> $O/rc.txt
MI355_ROWS_FINISH_OLD=1 timeout 900 python bench.py --config qwen3 --no-cpu-baseline > $O/qwen3_b64_oldfinish.json 2> $O/qwen3_b64_oldfinish.err; echo "qwen3 b64 oldfinish rc=$?" >> $O/rc.txt
timeout 900 python bench.py --config whisper --no-cpu-baseline > $O/whisper_new.json 2> $O/whisper_new.err; echo "whisper rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/bench.py --config qwen3 --no-cpu-baseline --steps 1 --warmup 1 > $O/prof_q.log 2>&1
DB=$(find $O/prof_q -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_qwen3_b64.txt 2>&1
rm -rf $O/prof_q
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_c.log 2>&1
DB=$(find $O/prof_c -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_csm.txt 2>&1
rm -rf $O/prof_c
cd $GRAFT_REPO_ROOT
cat $O/rc.txt; tail -5 $O/pytest_c12.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("csm_new","csm_fp8_new","qwen3_b1_new","qwen3_b64_new","qwen3_b64_oldfinish","whisper_new"):
    try:
        d=json.load(open(O+"/%s.json"%n)); print(n, round(d["value"],2), d["unit"], "ms/step", round(d.get("ms_per_step",0),3), "ms/frame", round(d.get("ms_per_frame",0),3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "ttfb", d.get("ttfb_ms"))
    except Exception as e: print(n, "ERR", e)
PY
head -16 $O/kstats_qwen3_b64.txt | cut -c1-60,120-230
head -16 $O/kstats_csm.txt | cut -c1-60,120-230
