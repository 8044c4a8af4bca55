#!/bin/bash
# round 6 call 6: precision 6 (fp16 hi + FP4 lo): conv parity against the scheme's numpy statement and the exact product, Kokoro at the mode's bars,
# per-shape A/B of the precisions, the contract line in modes 6 and 5 on the same box; + the codec-encode tests (re-synchronising walk)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 600 python -m pytest tests/test_conv_mx_gpu.py -x -q -s > $O/pytest_conv_mx.txt 2>&1; echo "pytest conv_mx rc=$?" >> $R
timeout 900 python -m pytest tests/test_kokoro_gpu.py -x -q -s -k "precision6 or precision5_mx" > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 400 python tools/bench_conv.py --prec-ab --batch 64 --out $O/conv_prec_ab_b64.txt > /dev/null 2> $O/conv_prec_ab.err; echo "prec ab rc=$?" >> $R
timeout 600 python bench.py --precision 6 --no-cpu-baseline --no-latency --no-pmc > $O/bench_p6.json 2> $O/bench_p6.err; echo "bench p6 rc=$?" >> $R
timeout 600 python bench.py --precision 5 --no-cpu-baseline --no-latency --no-pmc --no-secondary-precision > $O/bench_p5.json 2> $O/bench_p5.err; echo "bench p5 rc=$?" >> $R
timeout 900 python -m pytest tests/test_codec_encode_gpu.py tests/test_zz_margin_floors_gpu.py -q -s > $O/pytest_encode.txt 2>&1; echo "pytest encode rc=$?" >> $R
cat $R; grep -E "precision 6|passed|failed|Error|assert" $O/pytest_conv_mx.txt | tail -30 | cut -c1-200; grep -E "kokoro precision|passed|failed" $O/pytest_kokoro.txt | cut -c1-250
cat $O/conv_prec_ab_b64.txt | grep -E "p5|p6|p2" ; cut -c1-700 $O/bench_p6.json; echo; cut -c1-400 $O/bench_p5.json; echo
grep -E "forced \(oracle|margin rule|passed|failed" $O/pytest_encode.txt | cut -c1-230
