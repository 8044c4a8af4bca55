#!/bin/bash
# round 6 call 24: Whisper logits on the rows pipeline (A/B through MI355_WHISPER_LOGITS_ROWS), the tightened Kokoro envelope bar with its measured value
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 900 python -m pytest tests/test_whisper_gpu.py tests/test_reference_fixtures_gpu.py -x -q > $O/pytest_whisper.txt 2>&1; echo "pytest whisper rc=$?" >> $R
timeout 600 python -m pytest tests/test_kokoro_gpu.py -x -q -s -k "front_end_free_running or batch_equals_single" > $O/pytest_kokoro_s.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
for v in 1 0 1 0; do
  MI355_WHISPER_LOGITS_ROWS=$v timeout 400 python bench.py --config whisper 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('logits_rows=$v', d['value'], d['ms_per_step'], d['split_ms'])" >> $O/whisper_logits_ab.txt
done
cat $R; tail -3 $O/pytest_whisper.txt | cut -c1-200; grep -n "envelope\|passed\|failed" $O/pytest_kokoro_s.txt | cut -c1-200; cat $O/whisper_logits_ab.txt
