#!/bin/bash
# round 2 call 28: fp8 matrix-pipe GEMV v2 (tile loop + K chunks): tests, CSM at 8 sequences bf16 vs fp8
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py tests/test_codec_lm_gpu.py -q -m gpu > $O/t_fp8.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python tools/bench_csm.py --batch 8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b8_bf16.json 2> $O/b1.err; echo "b8 bf16 rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_csm.py --batch 8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline --weights fp8 > $O/bench_csm_b8_fp8.json 2> $O/b2.err; echo "b8 fp8 rc=$?" >> $O/rc.txt
MI355_GEMV_MFMA_FP8=0 timeout 600 python tools/bench_csm.py --batch 8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline --weights fp8 > $O/bench_csm_b8_fp8_fma.json 2> $O/b3.err; echo "b8 fp8 fma rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_f -o p -- python $GRAFT_REPO_ROOT/tools/bench_csm.py --batch 8 --frames 8 --steps 1 --warmup 1 --no-cpu-baseline --weights fp8 > $O/prof_f.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_f -name "*results.db" | head -1) 2 --by-grid > $O/kstats_csm_b8_fp8_bygrid.txt 2>&1
rm -rf $O/prof_f
cd $GRAFT_REPO_ROOT
tail -12 $O/t_fp8.log; cat $O/rc.txt; for f in bench_csm_b8_bf16 bench_csm_b8_fp8 bench_csm_b8_fp8_fma; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_frame'])"; done; head -8 $O/kstats_csm_b8_fp8_bygrid.txt | cut -c1-160
