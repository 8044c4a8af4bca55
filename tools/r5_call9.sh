#!/bin/bash
# round 5 call 9: v_cvt_scalef32 probe; STFT kernel with the next tile's samples prefetched + two mel accumulators; bench with the third PMC pass (mfma_busy_frac)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 60 tools/bin/cvt_scale_probe > $O/cvt_scale_probe.txt 2>&1; echo "probe rc=$?" >> $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "stft or logmel or fbank or fast" > $O/pytest_dsp.txt 2>&1; echo "pytest dsp rc=$?" >> $R
timeout 200 python -m pytest tests/test_whisper_gpu.py -x -q -k "log_mel or mel" > $O/pytest_whisper_mel.txt 2>&1; echo "pytest whisper mel rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --steps 20 > $O/bench_dsp_whisper.json 2> $O/bench_dsp.err; echo "bench dsp rc=$?" >> $R
timeout 120 python tools/bench_dsp.py --case qwen3 --steps 50 > $O/bench_dsp_qwen3.json 2>> $O/bench_dsp.err; echo "bench dsp qwen3 rc=$?" >> $R
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cat $R; cat $O/cvt_scale_probe.txt; tail -2 $O/pytest_dsp.txt | cut -c1-200; tail -2 $O/pytest_whisper_mel.txt | cut -c1-200
python - <<'PY'
import json
for f in ("bench_dsp_whisper", "bench_dsp_qwen3"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"]["kernel_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "err", d["max_abs_err_vs_oracle"])
    except Exception as e:
        print(f, "unreadable", e)
try:
    d = json.load(open("gpurun_out/bench_default.json"))
    r = d["roofline"]
    print("bench", d["value"], d["ms_per_step"], "p2", d.get("value_precision2"), "frac", r["frac"], "traffic", r["traffic"], "x alg", r["traffic_over_algorithmic"], "mfma_busy", r.get("mfma_busy_frac"), r.get("mfma_busy_frac_dominant_kernel"), "conv ms", r["conv_gemm_ms_per_step"])
except Exception as e:
    print("bench unreadable", e)
PY
grep "bench +" $O/bench_default.err | cut -c1-200
