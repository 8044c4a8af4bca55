#!/bin/bash
# round 6 call 1: where does conv_ws4_kernel<5, 2> wait?  (a) sanity of the edited kernel header, (b) timing ablations of the precision-5 kernel,
# (c) s_memtime timeline incl. barrier waits and the producer wave's split, (d) SQ / TCP / TA counters, (e) clock + power while it runs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python -m pytest tests/test_conv_mx_gpu.py -x -q > $O/pytest_conv_mx.txt 2>&1; echo "pytest conv_mx rc=$?" >> $R
timeout 300 python tools/bench_conv.py --ablate --precision 5 --batch 64 --out $O/conv_ablate_p5_b64.txt > /dev/null 2> $O/conv_ablate.err; echo "ablate rc=$?" >> $R
timeout 300 python tools/conv_timeline.py --precision 5 --batch 64 --out $O/conv_timeline_p5_b64.txt > /dev/null 2> $O/conv_timeline.err; echo "timeline rc=$?" >> $R
# clock / power while the k = 11 kernel runs back to back
( for i in $(seq 1 24); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction)" | tr '\n' ';'; echo; sleep 0.5; done ) > $O/smi_during_conv.txt &
SMI=$!
timeout 120 python tools/bench_conv.py --pmc5 --batch 64 --loop-seconds 8 --out $O/conv_pmc5_plain.txt > $O/conv_loop.txt 2>&1; echo "loop rc=$?" >> $R
wait $SMI
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum TD_TD_BUSY_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_CACHE_MISS_sum" \
            "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --pmc5 --batch 64 > /dev/null 2> $O/pmc_$i.err
  echo "pmc pass $i rc=$?" >> $R
  DB=$(find $O/pmc_$i -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" conv_ws4 >> $O/pmc_conv_p5_shapes.txt 2>/dev/null
  rm -rf $O/pmc_$i
done
cd "$GRAFT_REPO_ROOT"
cat $R; tail -3 $O/pytest_conv_mx.txt | cut -c1-200
cat $O/conv_ablate_p5_b64.txt; cat $O/conv_timeline_p5_b64.txt; tail -3 $O/conv_loop.txt; sed -n '4,8p;16,20p' $O/smi_during_conv.txt | cut -c1-400
cat $O/pmc_conv_p5_shapes.txt | cut -c1-150
