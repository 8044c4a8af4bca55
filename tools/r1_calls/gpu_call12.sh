#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for b in 64 128 256; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err
  echo "b=$b rc=$?"; cat gpurun_out/bench_b$b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'])"; tail -n 2 gpurun_out/bench_b$b.err
done
