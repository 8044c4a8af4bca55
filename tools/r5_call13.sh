#!/bin/bash
# round 5 call 13: the four new log-mel front ends on the GPU, dsp tests after the mode-4 / guard changes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 300 python -m pytest tests/test_frontends_gpu.py -x -q > $O/pytest_frontends.txt 2>&1; echo "pytest frontends rc=$?" >> $R
MI355_FFT_FAST=0 timeout 300 python -m pytest tests/test_frontends_gpu.py -x -q > $O/pytest_frontends_stockham.txt 2>&1; echo "pytest frontends (LDS Stockham kernel) rc=$?" >> $R
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_whisper_gpu.py tests/test_reference_fixtures_gpu.py -x -q -k "stft or logmel or fbank or fast or mel or dsp" > $O/pytest_dsp.txt 2>&1; echo "pytest dsp rc=$?" >> $R
cat $R; tail -12 $O/pytest_frontends.txt | cut -c1-250; tail -3 $O/pytest_frontends_stockham.txt | cut -c1-250; tail -2 $O/pytest_dsp.txt | cut -c1-200
