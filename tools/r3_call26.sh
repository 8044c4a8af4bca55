#!/bin/bash
# round 3 call 26: split rule (deep K loops up to 256 tiles): one utterance per call, EnCodec at one clip, default line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/bench_codecs.py --batch 1 --only encodec > $O/codecs_b1.jsonl 2> $O/codecs_b1.err; echo "codecs b1 rc=$?" > $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
d=json.load(open(O+"/bench_default.json")); print("default", round(d["value"]/1e6,2), "M ms/step", round(d["ms_per_step"],3), "roofline", round(d["roofline"]["frac"],4), "lat", d["latency_b1"]["ms"])
for l in open(O+"/codecs_b1.jsonl"):
    d=json.loads(l); print(d.get("config",{}).get("workload","?")[:50], round(d["value"]/1e6,2), "M samples/s", round(d["ms_per_step"],3), "ms")
PY
