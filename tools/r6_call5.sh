#!/bin/bash
# round 6 call 5: the whole GPU suite on the build with the column-wave conv layout / producer fast path / RVQ re-synchronisation hooks;
# the per-GPU lines of the 8-GPU configurations measured on one GPU; decode-line baselines before the launch-count work
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_full.txt 2>&1; echo "pytest full rc=$?" >> $R
timeout 300 python bench.py --batch 8 --no-pmc --no-cpu-baseline --steps 10 > $O/bench_kokoro_b8.json 2> $O/bench_kokoro_b8.err; echo "kokoro b8 rc=$?" >> $R
timeout 400 python bench.py --config qwen3 --batch 8 --no-cpu-baseline > $O/bench_qwen3_b8.json 2> $O/bench_qwen3_b8.err; echo "qwen3 b8 rc=$?" >> $R
timeout 400 python bench.py --config qwen3 --no-cpu-baseline > $O/bench_qwen3_b64.json 2> $O/bench_qwen3_b64.err; echo "qwen3 b64 rc=$?" >> $R
timeout 400 python bench.py --config csm --no-cpu-baseline > $O/bench_csm.json 2> $O/bench_csm.err; echo "csm rc=$?" >> $R
cat $R; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 | cut -c1-300; grep -E "margin rule|forced \(oracle" $O/pytest_full.txt | cut -c1-260
for f in kokoro_b8 qwen3_b8 qwen3_b64 csm; do echo "== $f"; cut -c1-1400 $O/bench_$f.json; done
