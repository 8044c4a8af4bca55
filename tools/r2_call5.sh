#!/bin/bash
# round-2 call 5: in-network A/B of the ws4 launch modes (persistent / one workgroup per tile) x consumer priority, same box, same process order
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/feat_ab5.txt
for rep in 1 2; do
for feat in 8 9 0 1; do
  MI355_CONV_WS_FEAT=$feat timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/bench5_f$feat.json 2> gpurun_out/bench5_f$feat.err
  python - "$feat" "$rep" <<'PY' >> gpurun_out/feat_ab5.txt
import json, sys
d = json.loads(open(f"gpurun_out/bench5_f{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("rep", sys.argv[2], "feat", sys.argv[1], "ms_per_step %.2f" % d["ms_per_step"], "conv_ms %.2f" % d["roofline"]["conv_gemm_ms_per_step"], "frac %.4f" % d["roofline"]["frac"])
PY
done
done
MI355_CONV_WS_VARIANT=7 timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/bench5_ws3.json 2> gpurun_out/bench5_ws3.err
python - <<'PY' >> gpurun_out/feat_ab5.txt
import json
d = json.loads(open("gpurun_out/bench5_ws3.json").read().strip().splitlines()[-1])
print("ws3", "ms_per_step %.2f" % d["ms_per_step"], "conv_ms %.2f" % d["roofline"]["conv_gemm_ms_per_step"], "frac %.4f" % d["roofline"]["frac"])
PY
cat gpurun_out/feat_ab5.txt
