#!/bin/bash
# round 3 call 27: same-box A/B of the split rule: EnCodec at one clip (per-kernel, split on / off), one utterance per call, twice each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2; do
  for v in 1 0; do
    MI355_CONV_SPLIT=$v timeout 600 python tools/bench_codecs.py --batch 1 --only encodec > $O/enc_s${v}_r$rep.jsonl 2>> $O/err.txt
    MI355_CONV_SPLIT=$v timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-roofline > $O/b1_s${v}_r$rep.json 2>> $O/err.txt
  done
done
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  MI355_CONV_SPLIT=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_e$v -o p -- python $GRAFT_REPO_ROOT/tools/bench_codecs.py --batch 1 --only encodec --steps 5 --warmup 2 > $O/prof_e$v.log 2>&1
  DB=$(find $O/prof_e$v -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 7 --by-grid > $O/kstats_encodec_b1_split$v.txt 2>&1
  rm -rf $O/prof_e$v
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for f in sorted(glob.glob(O+"/enc_s*_r*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(os.path.basename(f), round(d["ms_per_step"],3), "ms")
for f in sorted(glob.glob(O+"/b1_s*_r*.json")):
    d=json.load(open(f)); print(os.path.basename(f), "ms/step", round(d["ms_per_step"],3), "lat", round(d["latency_b1"]["ms"],3))
PY
for v in 1 0; do echo "== split $v"; head -14 $O/kstats_encodec_b1_split$v.txt | cut -c1-180; done
