#!/bin/bash
# round 2 call 46 (last 5.9 GPU-minutes): the device-side polyphase resampler (tests + timing), then as much of the full suite as the budget allows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 python -m pytest tests/test_resample_gpu.py -q -m gpu > $O/t_resample.log 2>&1; echo "resample rc=$?" > $O/rc46.txt
timeout 60 python tools/bench_resample.py > $O/bench_resample.json 2> $O/bench_resample.err; echo "bench_resample rc=$?" >> $O/rc46.txt
timeout 40 python tools/bench_resample.py --orig 24000 --target 16000 > $O/bench_resample_24k.json 2>> $O/bench_resample.err; echo "bench_resample24 rc=$?" >> $O/rc46.txt
timeout 30 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke46.log 2>&1; echo "smoke rc=$?" >> $O/rc46.txt
timeout 200 python -m pytest tests -q -x -m gpu --deselect tests/test_resample_gpu.py > $O/t_full46.log 2>&1; echo "full rc=$?" >> $O/rc46.txt
tail -5 $O/t_resample.log; cat $O/rc46.txt; cat $O/bench_resample.json $O/bench_resample_24k.json; tail -2 $O/smoke46.log; tail -3 $O/t_full46.log
