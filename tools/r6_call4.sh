#!/bin/bash
# round 6 call 4: producer trims (interior-window loads / unmasked conversion / two-instruction abs max) + same-box A/B of the two consumer layouts
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python -m pytest tests/test_conv_mx_gpu.py -x -q > $O/pytest_conv_mx.txt 2>&1; echo "pytest conv_mx rc=$?" >> $R
timeout 900 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 400 python tools/bench_conv.py --ablate --precision 5 --batch 64 --out $O/conv_ablate_p5_b64.txt > /dev/null 2> $O/conv_ablate.err; echo "ablate rc=$?" >> $R
timeout 600 python bench.py --no-cpu-baseline --no-latency > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; tail -3 $O/pytest_conv_mx.txt | cut -c1-300; grep -E "kokoro|passed|failed" $O/pytest_kokoro.txt | tail -8 | cut -c1-250
cat $O/conv_ablate_p5_b64.txt
cut -c1-1500 $O/bench_default.json; tail -2 $O/bench_default.err | cut -c1-300
