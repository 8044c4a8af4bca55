#!/bin/bash
# round 5 call 14: silent-tile shortcut of the STFT kernel: bit-identity test, every dsp / front-end test, the dsp line with its dense-input leg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_whisper_gpu.py tests/test_reference_fixtures_gpu.py tests/test_frontends_gpu.py -x -q -k "stft or logmel or fbank or fast or mel or dsp or front or silent" > $O/pytest_dsp.txt 2>&1; echo "pytest dsp rc=$?" >> $R
timeout 200 python -m pytest tests/test_whisper_gpu.py tests/test_qwen3_clone_gpu.py -x -q > $O/pytest_whisper_clone.txt 2>&1; echo "pytest whisper + qwen3 clone rc=$?" >> $R
timeout 120 python bench.py --config dsp --steps 20 > $O/bench_dsp_whisper.json 2> $O/bench_dsp.err; echo "bench dsp rc=$?" >> $R
cat $R; tail -3 $O/pytest_dsp.txt | cut -c1-250; tail -2 $O/pytest_whisper_clone.txt | cut -c1-200
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_dsp_whisper.json"))
print("dsp ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"]["kernel_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "err", d["max_abs_err_vs_oracle"], "dense", d.get("dense_input"))
PY
