#!/bin/bash
# round 2 call 29: K split over workgroups in the 16-bit matrix-pipe GEMV (down projections): tests, Qwen3 / Whisper / CSM-8 benches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_lm_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_whisper_gpu.py tests/test_tts_model_protocol_gpu.py -q -m gpu > $O/t_ks.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config qwen3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_qwen3.json 2> $O/bq.err; echo "qwen3 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --config whisper --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_whisper.json 2> $O/bw.err; echo "whisper rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_csm.py --batch 8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b8_bf16.json 2> $O/b1.err; echo "csm8 rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q -o p -- python $GRAFT_REPO_ROOT/bench.py --config qwen3 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_q.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_q -name "*results.db" | head -1) 2 --by-grid > $O/kstats_qwen3_bygrid.txt 2>&1
rm -rf $O/prof_q
cd $GRAFT_REPO_ROOT
tail -6 $O/t_ks.log; cat $O/rc.txt; for f in bench_qwen3 bench_csm_b8_bf16; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_frame'])"; done; python -c "
import json
d=json.loads(open('$O/bench_whisper.json').read().strip().splitlines()[-1]); print('whisper', d['value'], d['split_ms'])"; head -14 $O/kstats_qwen3_bygrid.txt | cut -c1-150
