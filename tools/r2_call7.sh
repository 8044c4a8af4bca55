#!/bin/bash
# round-2 call 7: drop-in surfaces of Qwen3-TTS / CSM (load_model -> generate vs the oracle), left-padded batches, RoPE bounds; full GPU suite
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests/test_tts_model_protocol_gpu.py -q --tb=short -p no:cacheprovider -x > gpurun_out/t_proto7.log 2>&1
echo "protocol tests rc=$?" | tee -a $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full7.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py > gpurun_out/bench_csm7.json 2> gpurun_out/bench_csm7.err; echo "csm rc=$?" | tee -a $R
cat $R; tail -n 40 gpurun_out/t_proto7.log | cut -c1-300; tail -n 25 gpurun_out/t_full7.log | cut -c1-300; cut -c1-600 gpurun_out/bench_csm7.json; tail -3 gpurun_out/bench_csm7.err
