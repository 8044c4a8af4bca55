#!/bin/bash
# round 5 call 1: (a) the new fast STFT / log-mel kernels: A/B test + existing dsp / whisper front-end tests + the dsp bench lines;
# (b) Kokoro parity in the new DEFAULT mode (5), including the 64-utterance test through shard.kokoro_step; (c) SQ counters of the conv kernels
# at the benchmarked batch (item 3 of VERDICT r4); (d) the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "stft or logmel or fbank or fast" > $O/pytest_dsp.txt 2>&1; echo "pytest dsp rc=$?" >> $R
timeout 300 python -m pytest tests/test_whisper_gpu.py -x -q -k "log_mel or mel" > $O/pytest_whisper_mel.txt 2>&1; echo "pytest whisper mel rc=$?" >> $R
timeout 300 python bench.py --config dsp --steps 20 --warmup 3 > $O/bench_dsp_whisper.json 2> $O/bench_dsp_whisper.err; echo "bench dsp rc=$?" >> $R
timeout 300 python tools/bench_dsp.py --ab --steps 20 > $O/bench_dsp_whisper_ab.json 2>> $O/bench_dsp_whisper.err; echo "bench dsp ab rc=$?" >> $R
timeout 300 python tools/bench_dsp.py --case qwen3 --ab --steps 50 > $O/bench_dsp_qwen3_ab.json 2>> $O/bench_dsp_whisper.err; echo "bench dsp qwen3 rc=$?" >> $R
timeout 900 python -m pytest tests/test_kokoro_gpu.py -x -q -s > $O/pytest_kokoro.txt 2>&1; echo "pytest kokoro rc=$?" >> $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $R
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$pass" | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-latency --no-secondary-precision > /dev/null 2> $O/pmc_$n.err
  echo "pmc $n rc=$?" >> $R
  DB=$(find $O/pmc_$n -name "*_results.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" conv_ >> $O/pmc_conv_b64.txt 2>/dev/null
  rm -rf $O/pmc_$n
done
cd "$GRAFT_REPO_ROOT"
cat $R; tail -3 $O/pytest_dsp.txt | cut -c1-300; tail -3 $O/pytest_whisper_mel.txt | cut -c1-300; tail -4 $O/pytest_kokoro.txt | cut -c1-300
cut -c1-900 $O/bench_dsp_whisper.json; cut -c1-1500 $O/bench_dsp_whisper_ab.json | tail -c 700; cut -c1-600 $O/bench_default.json; tail -2 $O/bench_default.err | cut -c1-300
