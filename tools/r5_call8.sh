#!/bin/bash
# round 5 call 8: branch-free producers (+ v_med3 / v_fma_mix in mode 5) of conv_ws4, the wave-independent STFT kernel, the guarded bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv_mx_gpu.py -x -q > $O/pytest_kernels.txt 2>&1; echo "pytest kernels rc=$?" >> $R
timeout 200 python -m pytest tests/test_whisper_gpu.py -x -q -k "log_mel or mel" > $O/pytest_whisper_mel.txt 2>&1; echo "pytest whisper mel rc=$?" >> $R
timeout 300 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --ab --steps 20 > $O/bench_dsp_whisper_ab.json 2> $O/bench_dsp.err; echo "bench dsp ab rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --case qwen3 --ab --steps 50 > $O/bench_dsp_qwen3_ab.json 2>> $O/bench_dsp.err; echo "bench dsp qwen3 rc=$?" >> $R
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; tail -3 $O/pytest_kernels.txt | cut -c1-300; tail -2 $O/pytest_whisper_mel.txt | cut -c1-200; grep -a "kokoro\|passed\|failed\|Error" $O/pytest_kokoro.txt | cut -c1-260 | tail -14
python - <<'PY'
import json
for f in ("bench_dsp_whisper_ab", "bench_dsp_qwen3_ab"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"]["kernel_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "prev", d.get("previous_kernel", {}).get("kernel_ms_per_step"), "err", d["max_abs_err_vs_oracle"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cut -c1-3500 $O/bench_default.json; grep "bench +" $O/bench_default.err | cut -c1-200
