#!/bin/bash
# round-2 call 6: ws4 as the only wave-specialised kernel (all precisions / prologue kinds / epilogue families): full GPU suite, the contract
# bench line, codec / Whisper / Qwen3 / CSM secondary lines
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_full6.log 2>&1
echo "full suite rc=$?" | tee -a $R
timeout 300 python bench.py --no-cpu-baseline --shape-table gpurun_out/shape_table6.txt > gpurun_out/bench6.json 2> gpurun_out/bench6.err
echo "bench rc=$?" | tee -a $R
timeout 300 python tools/bench_codecs.py > gpurun_out/bench_codecs6.jsonl 2> gpurun_out/bench_codecs6.err; echo "codecs rc=$?" | tee -a $R
timeout 240 python tools/bench_whisper.py --no-cpu-baseline > gpurun_out/bench_whisper6.json 2> gpurun_out/bench_whisper6.err; echo "whisper rc=$?" | tee -a $R
timeout 240 python tools/bench_qwen3.py --no-cpu-baseline > gpurun_out/bench_qwen36.json 2> gpurun_out/bench_qwen36.err; echo "qwen3 rc=$?" | tee -a $R
timeout 240 python tools/bench_csm.py --no-cpu-baseline > gpurun_out/bench_csm6.json 2> gpurun_out/bench_csm6.err; echo "csm rc=$?" | tee -a $R
cat $R; tail -n 12 gpurun_out/t_full6.log | cut -c1-250
cut -c1-900 gpurun_out/bench6.json; tail -n 2 gpurun_out/bench6.err
cut -c1-700 gpurun_out/bench_codecs6.jsonl; tail -n 3 gpurun_out/bench_codecs6.err
for n in whisper6 qwen36 csm6; do cut -c1-700 gpurun_out/bench_$n.json; tail -n 2 gpurun_out/bench_$n.err; done
