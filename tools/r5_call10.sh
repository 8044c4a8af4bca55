#!/bin/bash
# round 5 call 10: scaled fp8 conversion in the mode-5 producers; STFT kernel with the power rows on top of the transform buffer: 6 waves x 2 workgroups
# per CU (no prefetch) against 4 waves x 2 with the register prefetch (MI355_FFT_PREFETCH=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv_mx_gpu.py -x -q > $O/pytest_kernels.txt 2>&1; echo "pytest kernels rc=$?" >> $R
MI355_FFT_PREFETCH=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "stft or logmel or fbank or fast" > $O/pytest_dsp_prefetch.txt 2>&1; echo "pytest dsp (prefetch cfg) rc=$?" >> $R
timeout 200 python -m pytest tests/test_whisper_gpu.py -x -q -k "log_mel or mel" > $O/pytest_whisper_mel.txt 2>&1; echo "pytest whisper mel rc=$?" >> $R
timeout 300 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --steps 20 > $O/bench_dsp_whisper.json 2> $O/bench_dsp.err; echo "bench dsp rc=$?" >> $R
MI355_FFT_PREFETCH=1 timeout 120 python tools/bench_dsp.py --steps 20 --no-cpu-baseline > $O/bench_dsp_whisper_prefetch.json 2>> $O/bench_dsp.err; echo "bench dsp prefetch rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --case qwen3 --steps 50 > $O/bench_dsp_qwen3.json 2>> $O/bench_dsp.err; echo "bench dsp qwen3 rc=$?" >> $R
timeout 420 python bench.py --no-pmc > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; tail -2 $O/pytest_kernels.txt | cut -c1-200; tail -2 $O/pytest_dsp_prefetch.txt | cut -c1-200; grep -a "x 64\|x 4)\|passed\|failed\|Error" $O/pytest_kokoro.txt | cut -c1-260 | tail -6
python - <<'PY'
import json
for f in ("bench_dsp_whisper", "bench_dsp_whisper_prefetch", "bench_dsp_qwen3"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"]["kernel_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "err", d["max_abs_err_vs_oracle"])
    except Exception as e:
        print(f, "unreadable", e)
try:
    d = json.load(open("gpurun_out/bench_default.json"))
    r = d["roofline"]
    print("bench", d["value"], d["ms_per_step"], "p2", d.get("value_precision2"), "frac", r["frac"], "conv ms", r["conv_gemm_ms_per_step"], "check", d.get("batch_vs_single"))
except Exception as e:
    print("bench unreadable", e)
PY
