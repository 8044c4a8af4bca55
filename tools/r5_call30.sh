#!/bin/bash
# round 5 call 30: the final library (rvq.o and source.o differ from the library of call 25's full validation): the kernel-level suites and the files around
# the two changed objects once more, plus smoke()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$O/rc.txt; : > $R
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_kitten_gpu.py tests/test_conv_mx_gpu.py tests/test_edge_cases_gpu.py tests/test_interpolate_gpu.py tests/test_resample_gpu.py tests/test_frontends_gpu.py -q -m gpu > $O/pytest_final_subset.txt 2>&1; echo "pytest rc=$?" >> $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_final.txt 2>&1; echo "smoke rc=$?" >> $R
cat $R; tail -2 $O/pytest_final_subset.txt | cut -c1-200; tail -1 $O/smoke_final.txt
grep -E "^(FAILED|ERROR)|^E " $O/pytest_final_subset.txt | head -10 | cut -c1-300
