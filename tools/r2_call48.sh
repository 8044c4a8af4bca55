#!/bin/bash
# round 2 call 48: the four-outputs-per-thread resampler kernel: tests, then the three timing lines of call 47
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 60 python -m pytest tests/test_resample_gpu.py -q -m gpu > $O/t_resample48.log 2>&1; echo "resample rc=$?"
timeout 30 python tools/bench_resample.py > $O/bench_resample48.json 2> $O/bench_resample48.err; echo "rc=$?"
timeout 20 python tools/bench_resample.py --orig 24000 --target 16000 > $O/bench_resample48_24k.json 2>> $O/bench_resample48.err; echo "rc=$?"
timeout 20 python tools/bench_resample.py --orig 48000 --target 16000 --rows 8 > $O/bench_resample48_48k.json 2>> $O/bench_resample48.err; echo "rc=$?"
tail -4 $O/t_resample48.log; cat $O/bench_resample48.json $O/bench_resample48_24k.json $O/bench_resample48_48k.json
