#!/bin/bash
# round-1 call 35 (last of the round's GPU budget): fused RoPE epilogue (CSM / Mimi decode steps) and the streaming matrix-pipe GEMV (K > 2048):
# the complete suite at this commit, the GEMV tests again with the streaming variant forced on every norm-free call, then the decode lines A/B
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full35.log 2>&1
echo "full suite rc=$?" | tee -a $R
MI355_GEMV_MFMA_STREAM=2 timeout 120 python -m pytest tests/test_transformer_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemv" > gpurun_out/t_gemv35_stream2.log 2>&1
echo "gemv tests (stream=2) rc=$?" | tee -a $R
timeout 120 python tools/bench_csm.py > gpurun_out/bench_csm_35.json 2> gpurun_out/bench_csm_35.err; echo "csm rc=$?" | tee -a $R
MI355_GEMV_ROPE=0 timeout 120 python tools/bench_csm.py > gpurun_out/bench_csm_35_norope.json 2> gpurun_out/bench_csm_35_norope.err; echo "csm (separate rope) rc=$?" | tee -a $R
timeout 120 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen3_35.json 2> gpurun_out/bench_qwen3_35.err; echo "qwen3 rc=$?" | tee -a $R
cat $R; tail -n 15 gpurun_out/t_full35.log | cut -c1-250; tail -n 4 gpurun_out/t_gemv35_stream2.log | cut -c1-200
python - <<'PY'
import json
for n in ("csm_35", "csm_35_norope", "qwen3_35"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 2), d["unit"], {k: round(d[k], 3) for k in d if "ms" in k and not isinstance(d[k], dict)})
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-300:])
PY
