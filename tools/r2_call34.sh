#!/bin/bash
# round 2 call 34: full validation after KittenTTS / coarse_f32 / SNAC noise / reference-fixture tests; Kokoro + KittenTTS bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export MI355_MARGIN_REPORT=$O/margin_report.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/t_full.log 2>&1; echo "full rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_reference_fixtures_gpu.py tests/test_kitten_gpu.py -q -s > $O/t_fixtures.log 2>&1; echo "fixtures rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --config kitten > $O/bench_kitten.json 2> $O/bench_kitten.err; echo "kitten rc=$?" >> $O/rc.txt
tail -8 $O/t_full.log; grep -n "vs reference run\|passed\|failed" $O/t_fixtures.log | head -20; tail -2 $O/smoke.log; cat $O/rc.txt; head -c 300 $O/bench_default.json; echo; head -c 300 $O/bench_kitten.json
