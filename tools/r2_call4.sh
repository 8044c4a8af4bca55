#!/bin/bash
# round-2 call 4: timing ablations of the persistent ws4 kernel, shader clock from the probe, SQ / TCP / TCC counters of ws3 vs ws4
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
R=gpurun_out/rc.txt; : > $R
timeout 300 python tools/bench_conv.py --batch 32 --ablate --out gpurun_out/conv_ablate4_b32.txt > /dev/null 2> gpurun_out/conv_ablate4.err
echo "ablate rc=$?" | tee -a $R
timeout 300 python tools/conv_timeline.py --batch 32 --out gpurun_out/conv_timeline4_b32.txt > /dev/null 2> gpurun_out/conv_timeline4.err
echo "timeline rc=$?" | tee -a $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt" 2>&1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_IFETCH" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo "$pass" | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pass -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --batch 32 --quick --rounds 2 > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n.err"
  echo "pmc $n rc=$?" | tee -a "$GRAFT_REPO_ROOT/$R"
  DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n" -name "*_results.db" | head -1)
  [ -n "$DB" ] && python "$GRAFT_REPO_ROOT/tools/rocpd_pmc.py" "$DB" conv_ >> "$GRAFT_REPO_ROOT/gpurun_out/pmc_conv4.txt" 2>/dev/null
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n"
done
cd "$GRAFT_REPO_ROOT"
cat $R; cat gpurun_out/conv_ablate4_b32.txt; tail -3 gpurun_out/conv_ablate4.err; cat gpurun_out/conv_timeline4_b32.txt | grep -v "^   "; cat gpurun_out/pmc_conv4.txt
grep -c . gpurun_out/rocprof_counters.txt
