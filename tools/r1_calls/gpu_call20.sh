#!/bin/bash
# round-1 call 20: one-launch decode-step runner (mega_step.hip): parity vs the multi-launch runner / oracle, then the three decode benches A/B
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 300 python -m pytest tests/test_lm_kernels_gpu.py -m gpu -q --tb=short -x -k "fused or stack" -p no:cacheprovider > gpurun_out/t_fused20.log 2>&1
echo "fused tests rc=$?" | tee -a $R
if [ "$(tail -n 1 $R | grep -c 'rc=0')" = "1" ]; then
timeout 600 python -m pytest tests/test_whisper_gpu.py tests/test_codec_lm_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_models20.log 2>&1
echo "model tests rc=$?" | tee -a $R
for f in 1 0; do
  MI355_STEP_FUSED=$f timeout 300 python tools/bench_qwen3.py > gpurun_out/bench_qwen3_f$f.json 2> gpurun_out/bench_qwen3_f$f.err; echo "qwen3 f$f rc=$?" | tee -a $R
  MI355_STEP_FUSED=$f timeout 300 python tools/bench_csm.py > gpurun_out/bench_csm_f$f.json 2> gpurun_out/bench_csm_f$f.err; echo "csm f$f rc=$?" | tee -a $R
  MI355_STEP_FUSED=$f timeout 300 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_f$f.json 2> gpurun_out/bench_whisper_f$f.err; echo "whisper f$f rc=$?" | tee -a $R
done
fi
cat $R; tail -n 25 gpurun_out/t_fused20.log | cut -c1-300; tail -n 8 gpurun_out/t_models20.log 2>/dev/null | cut -c1-300
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*_f[01].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), d.get("unit"), {k: d[k] for k in d if "ms" in k and not isinstance(d[k], dict)})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
