#!/bin/bash
# round 5 call 20: measurement lines of the codec ENCODE sides (tools/bench_codecs.py --encode) with a kernel trace, and the last number the fp8 decision
# needs: CSM-1B at 8 sequences on fp8 images with the fp8 matrix-pipe GEMV switched off everywhere (MI355_GEMV_MFMA_FP8=0: what deleting the kernel does)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python tools/bench_codecs.py --encode --batch 16 --seconds 10 --steps 5 --warmup 2 > $O/bench_codecs_encode.jsonl 2> $O/bench_codecs_encode.err; echo "bench encode rc=$?" >> $R
MI355_GEMV_MFMA_FP8=0 timeout 150 python tools/bench_csm.py --batch 8 --weights fp8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b8_fp8_nomfma.json 2> $O/bench_csm_b8_fp8_nomfma.err; echo "csm b8 fp8 (no fp8 mfma gemv) rc=$?" >> $R
timeout 150 python tools/bench_csm.py --batch 8 --weights fp8 --frames 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_csm_b8_fp8_again.json 2> $O/bench_csm_b8_fp8_again.err; echo "csm b8 fp8 (default, same box) rc=$?" >> $R
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_e -o p -- python $GRAFT_REPO_ROOT/tools/bench_codecs.py --encode --batch 16 --seconds 10 --steps 2 --warmup 1 > $O/prof_e.log 2>&1; echo "trace encode rc=$?" >> $R
DB=$(find $O/prof_e -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 3 > $O/kstats_codecs_encode.txt 2>&1
rm -rf $O/prof_e
cd $GRAFT_REPO_ROOT
cat $R
tail -3 $O/bench_codecs_encode.err | cut -c1-300
python - <<'PY'
import json
for ln in open("gpurun_out/bench_codecs_encode.jsonl"):
    try:
        d = json.loads(ln); r = d["roofline"] or {}
        print(d["metric"][34:60], round(d["value"] / 1e6, 1), "M samples/s", round(d["ms_per_step"], 2), "ms", round(d["x_realtime"]), "x rt; conv frac", round(r.get("frac", 0), 3), "hbm", round(r.get("hbm_view", {}).get("frac", 0), 3), "launches", r.get("launches"), "conv ms", round(r.get("conv_gemm_ms", 0), 2))
    except Exception as e:
        print("ERR", e, ln[:100])
for f in ("b8_fp8_nomfma", "b8_fp8_again"):
    try:
        d = json.load(open(f"gpurun_out/bench_csm_{f}.json")); print(f, round(d["ms_per_frame"], 3), "ms/frame")
    except Exception as e:
        print(f, "ERR", e)
PY
head -14 $O/kstats_codecs_encode.txt | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-170
