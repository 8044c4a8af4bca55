#!/bin/bash
# round-1 call 31: measurement lines of the widened codec rows (Vocos / DAC / SNAC decode) + kernel stats of the SNAC pass
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/bench_codecs.py > gpurun_out/bench_codecs_31.jsonl 2> gpurun_out/bench_codecs_31.err; echo "codecs rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_s" -o s -- python "$GRAFT_REPO_ROOT/tools/bench_codecs.py" --only snac --steps 3 --warmup 1 > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_s.err"
echo "rocprof snac rc=$?"
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/prof_s -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" 5 | cut -c1-200 > gpurun_out/snac_kernel_stats_31.txt 2>&1; rm -rf gpurun_out/prof_s
tail -n 3 gpurun_out/bench_codecs_31.err
python - <<'PY'
import json
for l in open("gpurun_out/bench_codecs_31.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["metric"][:60], "|", round(d["value"] / 1e6, 1), "M samples/s", round(d["x_realtime"]), "x RT", round(d["ms_per_step"], 2), "ms (wall", round(d["wall_ms_per_step"], 2), ") conv", round(r["conv_gemm_ms"], 2), "ms", round(r["achieved"], 1), "TF/s frac", round(r["frac"], 3), "hbm", round(r["hbm_view"]["achieved_GBps"]), "GB/s")
PY
head -n 12 gpurun_out/snac_kernel_stats_31.txt | cut -c1-150
