#!/bin/bash
# round 2 call 11: M=1 GEMV kernels + register-resident whisper step + free-running vocoder test; CSM / whisper / qwen3 bench + by-grid stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_whisper_gpu.py tests/test_tts_model_protocol_gpu.py -q -m gpu > $O/t_gemv.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_kokoro_gpu.py -q -m gpu -s > $O/t_kokoro.log 2>&1; echo "kokoro rc=$?" >> $O/rc.txt
for c in csm whisper qwen3; do
  timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
MI355_GEMV_M1=0 timeout 600 python bench.py --config csm --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_oldm1.json 2> $O/bench_csm_oldm1.err
timeout 600 python tools/bench_csm.py --weights fp8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_fp8.json 2> $O/bench_csm_fp8.err
cd /tmp; export TMPDIR=/tmp
for c in csm whisper; do
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_$c.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_$c -name "*results.db" | head -1) 2 --by-grid > $O/kstats_${c}_bygrid.txt 2>&1
rm -rf $O/prof_$c
done
cd $GRAFT_REPO_ROOT
tail -5 $O/t_gemv.log; grep -n "free-running\|passed\|failed" $O/t_kokoro.log | tail -5; cat $O/rc.txt
for f in bench_csm bench_csm_oldm1 bench_csm_fp8 bench_whisper bench_qwen3; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d.get('ms_per_frame'), d.get('split_ms'))"; done
