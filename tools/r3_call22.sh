#!/bin/bash
# round 3 call 22: secondary lines after split-K / 64-column tiles / folded quantiser (do the small-launch convs of Whisper, Mimi, the codecs move?), default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for c in whisper csm qwen3; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
MI355_CONV_SPLIT=0 timeout 900 python bench.py --config whisper --no-cpu-baseline > $O/bench_whisper_nosplit.json 2> $O/bench_whisper_nosplit.err; echo "whisper nosplit rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_codecs.py --batch 1 > $O/codecs_b1.jsonl 2> $O/codecs_b1.err; echo "codecs b1 rc=$?" >> $O/rc.txt
MI355_CONV_SPLIT=0 timeout 600 python tools/bench_codecs.py --batch 1 > $O/codecs_b1_nosplit.jsonl 2> $O/codecs_b1_nosplit.err; echo "codecs b1 nosplit rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for n in ("whisper","whisper_nosplit","csm","qwen3","default"):
    try:
        d=json.load(open(O+"/bench_%s.json"%n)); print(n, round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "roofline", round((d.get("roofline") or {}).get("frac",0),4), "lat", (d.get("latency_b1") or {}).get("ms"))
    except Exception as e: print(n, "ERR", e, open(O+"/bench_%s.err"%n).read()[-300:])
for n in ("codecs_b1","codecs_b1_nosplit"):
    try:
        for l in open(O+"/%s.jsonl"%n):
            d=json.loads(l); print(n, d.get("config",{}).get("workload","?")[:50], round(d["value"]/1e6,2), "M samples/s", round(d["ms_per_step"],3), "ms")
    except Exception as e: print(n, "ERR", e)
PY
