#!/bin/bash
# round 6 call 18: contract line with the per-shape table of the conv launches (which shapes are left outside the precision-6 kernel), Whisper line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 600 python bench.py --shape-table $O/shape_table_b64.txt --no-cpu-baseline --no-latency --no-secondary-precision > $O/bench_shape.json 2> $O/bench_shape.err; echo "bench rc=$?" >> $R
timeout 600 python bench.py --config whisper > $O/bench_whisper.json 2> $O/bench_whisper.err; echo "bench whisper rc=$?" >> $R
cat $R; cut -c1-400 $O/bench_shape.json; echo; cat $O/shape_table_b64.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_whisper.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["split_ms"], d["phase_rooflines"]["encoder"]["frac"])
PY
