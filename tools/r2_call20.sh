#!/bin/bash
# round 2 call 20: embedding lookup fused into the projection GEMV (CSM, one sequence)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_codec_lm_gpu.py tests/test_tts_model_protocol_gpu.py -q -m gpu -k "gemv or csm" > $O/t_gather.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --config csm --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_csm.json 2> $O/bench_csm.err; echo "csm rc=$?" >> $O/rc.txt
tail -4 $O/t_gather.log; cat $O/rc.txt; python -c "
import json
d=json.loads(open('$O/bench_csm.json').read().strip().splitlines()[-1]); print('csm', d['value'], d['ms_per_frame'])"
