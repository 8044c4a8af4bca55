#!/bin/bash
# round 2 call 47: timing of the device-side resampler (call 46's bench lines died on an import path)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 60 python tools/bench_resample.py > $O/bench_resample.json 2> $O/bench_resample.err; echo "rc=$?"
timeout 30 python tools/bench_resample.py --orig 24000 --target 16000 > $O/bench_resample_24k.json 2>> $O/bench_resample.err; echo "rc=$?"
timeout 30 python tools/bench_resample.py --orig 48000 --target 16000 --rows 8 > $O/bench_resample_48k.json 2>> $O/bench_resample.err; echo "rc=$?"
cat $O/bench_resample.json $O/bench_resample_24k.json $O/bench_resample_48k.json; tail -3 $O/bench_resample.err
