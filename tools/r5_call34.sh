#!/bin/bash
# round 5 call 34: the config[3] / config[4] lines on the FINAL build (Qwen3-TTS-1.7B at 64 utterances, CSM-1B at one sequence)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
for c in qwen3 csm; do
  timeout 100 python bench.py --config $c --no-cpu-baseline > $O/bench_final_$c.json 2> $O/bench_final_$c.err; echo "$c rc=$?" >> $R
done
cat $R
python - <<'PY'
import json
for c in ("qwen3", "csm"):
    try:
        d = json.loads(open(f"gpurun_out/bench_final_{c}.json").read().strip().splitlines()[-1]); print(c, round(d["value"], 1), d["unit"], round(d["ms_per_frame"], 3), "ms/frame frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(c, "ERR", e)
PY
