#!/bin/bash
# round-1 call 21: (1) the one-launch decode-step runner (mega_step.hip) and the fp8 weight images against the multi-launch runner / oracle,
# (2) the complete GPU suite in one process (what the driver runs) + smoke, with the step runner that survived (1),
# (3) decode benches A/B: fused vs multi-launch (Qwen3-TTS-1.7B, CSM-1B, Whisper-small), CSM with fp8 weight images
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
DESEL=""
timeout 420 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider \
  -k "fused or fp8 or stack_prefill" > gpurun_out/t_new21.log 2>&1
rc=$?; echo "new tests rc=$rc" | tee -a $R
if [ $rc -ne 0 ]; then
  # keep the verified multi-launch runner for everything below; the fused tests are then expected to fail and are left out
  export MI355_STEP_FUSED=0
  DESEL='not fused_step_runner and not (fp8_weight_images and True)'
  echo "FUSED RUNNER DISABLED for the rest of this call" | tee -a $R
  timeout 300 python -m pytest tests/test_lm_kernels_gpu.py tests/test_transformer_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "fp8 and not (fp8_weight_images and True)" > gpurun_out/t_fp8_multi21.log 2>&1
  echo "fp8 tests with the multi-launch runner rc=$?" | tee -a $R
fi
if [ -n "$DESEL" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$DESEL" > gpurun_out/t_full21.log 2>&1
else
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full21.log 2>&1
fi
echo "full suite rc=$? (MI355_STEP_FUSED=${MI355_STEP_FUSED:-1})" | tee -a $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke21.log 2>&1
echo "smoke rc=$?" | tee -a $R
for f in 1 0; do
  if [ "$f" = "1" ] && [ "${MI355_STEP_FUSED:-1}" = "0" ]; then continue; fi
  MI355_STEP_FUSED=$f timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_f$f.json 2> gpurun_out/bench_qwen3_f$f.err; echo "qwen3 f$f rc=$?" | tee -a $R
  MI355_STEP_FUSED=$f timeout 240 python tools/bench_csm.py > gpurun_out/bench_csm_f$f.json 2> gpurun_out/bench_csm_f$f.err; echo "csm f$f rc=$?" | tee -a $R
  MI355_STEP_FUSED=$f timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper_f$f.json 2> gpurun_out/bench_whisper_f$f.err; echo "whisper f$f rc=$?" | tee -a $R
  MI355_STEP_FUSED=$f timeout 240 python tools/bench_csm.py --weights fp8 > gpurun_out/bench_csm_fp8_f$f.json 2> gpurun_out/bench_csm_fp8_f$f.err; echo "csm fp8 f$f rc=$?" | tee -a $R
done
cat $R; tail -n 30 gpurun_out/t_new21.log | cut -c1-250; tail -n 12 gpurun_out/t_full21.log | cut -c1-250; tail -n 2 gpurun_out/smoke21.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*_f[01].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), d.get("unit"), {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)}, "frac", round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
