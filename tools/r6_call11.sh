#!/bin/bash
# round 6 call 11: what bounds the GEMM mode of conv_ws4 on the Whisper encoder's linears (96 000 rows)?  precisions 2 / 3 / 4, timing ablations
# (no weight loads / idle producers), SQ + TCC counters of the same launches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python tools/bench_conv.py --big-gemm --batch 64 --rounds 5 --out $O/conv_big_gemm_b64.txt > /dev/null 2> $O/conv_big_gemm.err; echo "big-gemm rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --big-gemm --batch 64 --rounds 2 > /dev/null 2> $O/pmc_$i.err
  echo "pmc pass $i rc=$?" >> $R
  DB=$(find $O/pmc_$i -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" conv_ws4 >> $O/pmc_big_gemm.txt 2>/dev/null
  rm -rf $O/pmc_$i
done
cd "$GRAFT_REPO_ROOT"
cat $R; cat $O/conv_big_gemm_b64.txt; tail -5 $O/conv_big_gemm.err; cat $O/pmc_big_gemm.txt | cut -c1-180
