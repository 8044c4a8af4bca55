#!/bin/bash
# round 5 call 12: full-depth stack parity (measurement for its bar), STFT kernel with v_log_f32, clean kernel trace of the contract step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 600 python -m pytest tests/test_lm_kernels_gpu.py -x -q -s -k "full_depth" > $O/pytest_full_depth.txt 2>&1; echo "pytest full depth rc=$?" >> $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "stft or logmel or fbank or fast" > $O/pytest_dsp.txt 2>&1; echo "pytest dsp rc=$?" >> $R
timeout 200 python -m pytest tests/test_whisper_gpu.py -x -q > $O/pytest_whisper.txt 2>&1; echo "pytest whisper rc=$?" >> $R
timeout 120 python bench.py --config dsp --steps 20 > $O/bench_dsp_whisper.json 2> $O/bench_dsp.err; echo "bench dsp rc=$?" >> $R
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-pmc --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check > $O/prof_k.log 2>&1
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $R; grep -a "full depth\|passed\|failed\|Error" $O/pytest_full_depth.txt | cut -c1-300; tail -2 $O/pytest_dsp.txt | cut -c1-200; tail -2 $O/pytest_whisper.txt | cut -c1-200
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_dsp_whisper.json"))
print("dsp ms/step", round(d["ms_per_step"], 4), "kernel ms", round(d["roofline"]["kernel_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "err", d["max_abs_err_vs_oracle"])
PY
head -14 $O/kstats_b64.txt | cut -c1-170
