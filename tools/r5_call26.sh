#!/bin/bash
# round 5 call 26: mi355_rvq_encode with several frames per workgroup: the bit-identity test of the forms, every test that searches codebooks (Mimi encode,
# Qwen3 tokenizer encode, codec encode sides, reference fixtures), the encode lines again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 420 python -m pytest tests/test_codec_encode_gpu.py tests/test_mimi_gpu.py tests/test_reference_fixtures_gpu.py tests/test_qwen3_clone_gpu.py -q -m gpu > $O/pytest_rvq_rows.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 300 python tools/bench_codecs.py --encode --batch 16 --seconds 10 --steps 5 --warmup 2 > $O/bench_codecs_encode2.jsonl 2> $O/bench_codecs_encode2.err; echo "bench encode rc=$?" >> $R
MI355_RVQ_ROWS=1 timeout 300 python tools/bench_codecs.py --encode --only encodec --batch 16 --seconds 10 --steps 5 --warmup 2 > $O/bench_encodec_encode_rvq1.json 2> $O/bench_encodec_encode_rvq1.err; echo "bench encodec rvq rows=1 rc=$?" >> $R
cat $R
tail -3 $O/pytest_rvq_rows.txt | cut -c1-250
grep -E "^(FAILED|ERROR)|Error|assert |^E " $O/pytest_rvq_rows.txt | head -20 | cut -c1-300
python - <<'PY'
import json
for f in ("bench_codecs_encode2.jsonl", "bench_encodec_encode_rvq1.json"):
    for ln in open("gpurun_out/" + f):
        try:
            d = json.loads(ln); print(f[:28], d["metric"][34:52], round(d["value"] / 1e6, 1), "M samples/s", round(d["ms_per_step"], 2), "ms")
        except Exception as e:
            print("ERR", e)
PY
