#!/bin/bash
# round 5 call 27: after the frames-per-workgroup search kernel (the only change since the full validation of call 25): the remaining test files that can reach
# mi355_rvq_encode, and a kernel trace of the contract step on the final build (rocprofv3 --kernel-trace --stats of the default bench command's timed region)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 420 python -m pytest tests/test_tts_model_protocol_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py tests/test_encodec_gpu.py tests/test_codec_lm_gpu.py -q -m gpu > $O/pytest_rest.txt 2>&1; echo "pytest rc=$?" >> $R
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-pmc --no-cpu-baseline --no-latency --no-secondary-precision --no-batch-check > $O/prof_k.log 2>&1; echo "trace rc=$?" >> $R
DB=$(find $O/prof_k -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 8 > $O/kstats_b64_final.txt 2>&1
rm -rf $O/prof_k
cd $GRAFT_REPO_ROOT
cat $R; tail -2 $O/pytest_rest.txt | cut -c1-200
grep -E "^(FAILED|ERROR)|^E " $O/pytest_rest.txt | head -10 | cut -c1-300
head -10 $O/kstats_b64_final.txt | cut -c1-170
