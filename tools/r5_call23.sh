#!/bin/bash
# round 5 call 23: KittenTTS extrema from the producing conv's epilogue (ABI 33): the kernel test, the KittenTTS suite, the quantised line with the
# partials and without (MI355_EXT_PARTIALS=0), same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 420 python -m pytest tests/test_kitten_gpu.py -q -m gpu -x > $O/pytest_kitten_ext.txt 2>&1; echo "pytest rc=$?" >> $R
for mode in 1 0; do
  MI355_EXT_PARTIALS=$mode timeout 240 python tools/bench_kitten.py > $O/bench_kitten_ext$mode.json 2> $O/bench_kitten_ext$mode.err; echo "kitten ext=$mode rc=$?" >> $R
done
cat $R
tail -4 $O/pytest_kitten_ext.txt | cut -c1-300
grep -E "^(FAILED|ERROR)|Error|assert |^E " $O/pytest_kitten_ext.txt | head -30 | cut -c1-300
python - <<'PY'
import json
for f in ("ext1", "ext0"):
    try:
        d = json.loads(open(f"gpurun_out/bench_kitten_{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"] / 1e6, 1), "M samples/s", round(d["ms_per_step"], 2), "ms; plain", round(d["without_activation_quant"]["ms_per_step"], 2), "ms")
    except Exception as e:
        print(f, "ERR", e)
PY
