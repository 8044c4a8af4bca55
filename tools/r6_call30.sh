#!/bin/bash
# round 6 call 30: the secondary lines on the final tree (Kokoro at 8 utterances per GPU, Qwen3-TTS at 64 and 8 utterances, CSM-1B)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python bench.py --batch 8 --no-pmc --no-cpu-baseline --steps 10 > $O/bench_kokoro_b8.json 2> $O/bench_kokoro_b8.err; echo "kokoro b8 rc=$?" >> $R
timeout 400 python tools/bench_qwen3.py --batch 8 --no-cpu-baseline > $O/bench_qwen3_b8.json 2> $O/bench_qwen3_b8.err; echo "qwen3 b8 rc=$?" >> $R
timeout 400 python bench.py --config qwen3 --no-cpu-baseline > $O/bench_qwen3_b64.json 2> $O/bench_qwen3_b64.err; echo "qwen3 b64 rc=$?" >> $R
timeout 400 python bench.py --config csm --no-cpu-baseline > $O/bench_csm.json 2> $O/bench_csm.err; echo "csm rc=$?" >> $R
cat $R
python - <<'PY'
import json
for f in ("kokoro_b8", "qwen3_b8", "qwen3_b64", "csm"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_frame"), d.get("split_ms"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "failed", e)
PY
