#!/bin/bash
# round 5 call 15: evidence runs -- HBM traffic counters of the STFT kernel, SQ counters of the conv kernels after the producer changes, the contract line at other batch sizes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "silent" > $O/pytest_silent.txt 2>&1; echo "pytest silent rc=$?" >> $R
for b in 8 256; do
  timeout 300 python bench.py --batch $b --no-pmc --no-cpu-baseline --no-latency --steps 10 > $O/bench_b$b.json 2> $O/bench_b$b.err; echo "bench b$b rc=$?" >> $R
done
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_dsp_$ctr -o p -- python $GRAFT_REPO_ROOT/tools/bench_dsp.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_dsp_$ctr.err; echo "pmc dsp $ctr rc=$?" >> $R
done
F=$(find $O/pmc_dsp_FETCH_SIZE -name "*_results.db" | head -1); W=$(find $O/pmc_dsp_WRITE_SIZE -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py "$F" "$W" stft_fast > $O/pmc_traffic_dsp.json 2>/dev/null; python $GRAFT_REPO_ROOT/tools/pmc_traffic.py "$F" "$W" logmel_finish > $O/pmc_traffic_dsp_finish.json 2>/dev/null
rm -rf $O/pmc_dsp_FETCH_SIZE $O/pmc_dsp_WRITE_SIZE
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo "$pass" | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-latency --no-secondary-precision > /dev/null 2> $O/pmc_$n.err
  echo "pmc $n rc=$?" >> $R
  DB=$(find $O/pmc_$n -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" conv_ws4 >> $O/pmc_conv_b64_after.txt 2>/dev/null
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" lstm_oct >> $O/pmc_conv_b64_after.txt 2>/dev/null
  rm -rf $O/pmc_$n
done
cd $GRAFT_REPO_ROOT
cat $R; tail -2 $O/pytest_silent.txt | cut -c1-200
python - <<'PY'
import json
for b in (8, 256):
    try:
        d = json.load(open(f"gpurun_out/bench_b{b}.json")); print("batch", b, round(d["value"] / 1e6, 2), "M samples/s", round(d["ms_per_step"], 2), "ms/step p2", round(d.get("value_precision2", 0) / 1e6, 2), "frac", round(d["roofline"]["frac"], 4))
    except Exception as e:
        print("batch", b, "ERR", e)
for f in ("pmc_traffic_dsp", "pmc_traffic_dsp_finish"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); print(f, {k[:60]: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in d["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
grep -A8 "conv_ws4_kernel<5, 2\|lstm_oct" $O/pmc_conv_b64_after.txt | cut -c1-150 | head -60
