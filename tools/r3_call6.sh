#!/bin/bash
# round 3 call 6: after the T = 2 / fused SwiGLU / unrolled attention slab loads changes: parity, Qwen3 at 64 utterances with 48 frames, kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "rows or tile_image or decode_attention" > $O/t_pipe.log 2>&1; echo "pipe rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -q -m gpu -k "real_widths or qwen3 or csm" > $O/t_real.log 2>&1; echo "real rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_qwen3.py --batch 64 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b64.json 2> $O/qwen3_b64.err; echo "q64 rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_qwen3.py --batch 32 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b32.json 2> $O/qwen3_b32.err; echo "q32 rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_qwen3.py --batch 16 --frames 48 --steps 1 --no-cpu-baseline > $O/qwen3_b16.json 2> $O/qwen3_b16.err; echo "q16 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/tools/bench_qwen3.py --batch 64 --frames 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_q.log 2>&1
DB=$(find $O/prof_q -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2 --by-grid > $O/kstats_qwen3_b64.txt 2>&1
rm -rf $O/prof_q
cd $GRAFT_REPO_ROOT
tail -5 $O/t_pipe.log; tail -5 $O/t_real.log; cat $O/rc.txt; python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for b in (64,32,16):
    d=json.load(open(O+"/qwen3_b%d.json"%b)); print('qwen3 b%d'%b, round(d['value'],1), d['split_ms'], round(d['ms_per_frame'],3), round(d['roofline']['frac'],4))
PY
head -16 $O/kstats_qwen3_b64.txt | cut -c1-200
