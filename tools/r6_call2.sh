#!/bin/bash
# round 6 call 2: the column-wave consumer layout of conv_ws4 precision 5 (half the weight traffic through L1): parity, bitwise tests, ablations, contract line;
# + the FP4 operand / conversion probe and the fp4 rate of the matrix pipe under the power cap
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 60 tools/bin/mfma_fp4_probe > $O/mfma_fp4_probe.jsonl 2>&1; echo "fp4 probe rc=$?" >> $R
MI355_MFMA_ONLY=35 timeout 120 tools/bin/mfma_peak 50 > $O/mfma_peak_fp4.jsonl 2>&1; echo "mfma peak rc=$?" >> $R
timeout 400 python -m pytest tests/test_conv_mx_gpu.py -x -q > $O/pytest_conv_mx.txt 2>&1; echo "pytest conv_mx rc=$?" >> $R
timeout 900 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 300 python tools/bench_conv.py --ablate --precision 5 --batch 64 --out $O/conv_ablate_p5_b64.txt > /dev/null 2> $O/conv_ablate.err; echo "ablate rc=$?" >> $R
timeout 300 python tools/conv_timeline.py --precision 5 --batch 64 --out $O/conv_timeline_p5_b64.txt > /dev/null 2> $O/conv_timeline.err; echo "timeline rc=$?" >> $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; tail -3 $O/pytest_conv_mx.txt | cut -c1-300; grep -E "kokoro|passed|failed" $O/pytest_kokoro.txt | tail -8 | cut -c1-250
cat $O/conv_ablate_p5_b64.txt; grep -E "^##|^8 tiles|producer|^    [1-3]" $O/conv_timeline_p5_b64.txt | cut -c1-330
cut -c1-1800 $O/bench_default.json; tail -2 $O/bench_default.err | cut -c1-300
grep '"A"' $O/mfma_fp4_probe.jsonl; grep '"B"' $O/mfma_fp4_probe.jsonl | head -34; cat $O/mfma_peak_fp4.jsonl
