#!/bin/bash
# round 3 call 4: rows pipeline v2 (attention reads slabs / writes planes, column-parallel row epilogue): parity, stage timings, T / workgroup sweep, Qwen3 at 64
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_transformer_kernels_gpu.py -q -m gpu -k "rows or tile_image or decode_attention or matrix_pipe" > $O/t_pipe.log 2>&1; echo "pipe rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py -q -m gpu -k "real_widths or qwen3 or prefill_and_decode or csm" > $O/t_real.log 2>&1; echo "real rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_gemv.py --rows 64 --pipe --iters 100 --tag pipe64 > $O/pipe_rows64.txt 2>&1
: > $O/pipe_sweep.txt
for t in 2 4; do for w in 384 512 768 1024; do
  MI355_ROWS_T=$t MI355_ROWS_WGS=$w timeout 120 python tools/bench_gemv.py --rows 64 --pipe --iters 100 --tag "T${t}w${w}" 2>&1 | grep -v "^{" | grep -v amdgpu | grep -v whisper >> $O/pipe_sweep.txt
done; done
timeout 600 python tools/bench_qwen3.py --batch 64 --frames 24 --steps 1 --no-cpu-baseline > $O/qwen3_b64.json 2> $O/qwen3_b64.err; echo "q64 rc=$?" >> $O/rc.txt
tail -5 $O/t_pipe.log; tail -5 $O/t_real.log; cat $O/rc.txt; grep -v "^{" $O/pipe_rows64.txt | grep -v amdgpu; python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/qwen3_b64.json")); print('qwen3 b64', d['value'], d['split_ms'], d['ms_per_frame'], d['roofline']['frac'])
# sweep table: gemm us per shape per config
rows={}
for l in open(O+"/pipe_sweep.txt"):
    f=l.split()
    if len(f)<10: continue
    tag=f[0]; shape=" ".join(f[1:3]); i=f.index("gemm"); rows.setdefault(shape,{})[tag]=float(f[i+1])
tags=sorted({t for r in rows.values() for t in r})
print("shape".ljust(20)," ".join(t.rjust(8) for t in tags))
for sh,r in rows.items(): print(sh.ljust(20)," ".join(("%8.2f"%r.get(t,-1)) for t in tags))
PY
tail -3 $O/qwen3_b64.err
