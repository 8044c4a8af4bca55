#!/bin/bash
# round 2 call 10: side kernels (sine / stft_magphase / istft_head) parity + timings, bench after the host-overhead fixes, CSM by-grid kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kokoro_gpu.py tests/test_api_gpu.py -q -m gpu -x > $O/t_side.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_kokoro2.json 2> $O/bench_kokoro2.err; echo "bench rc=$?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-roofline > $O/prof_k.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_k -name "*results.db" | head -1) 3 > $O/kstats_kokoro.txt 2>&1
rm -rf $O/prof_k
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_csm -o p -- python $GRAFT_REPO_ROOT/bench.py --config csm --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_csm.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/prof_csm -name "*results.db" | head -1) 2 --by-grid > $O/kstats_csm_bygrid.txt 2>&1
rm -rf $O/prof_csm
cd $GRAFT_REPO_ROOT
tail -3 $O/t_side.log; cat $O/rc.txt; head -c 300 $O/bench_kokoro2.json; echo; grep -n "sine\|stft_mag\|istft_head" $O/kstats_kokoro.txt
