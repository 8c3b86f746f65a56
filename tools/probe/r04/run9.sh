#!/bin/bash
# split accumulate (assignment-only sweep + k_accum_ranges): parity, then the ladder with the threshold at 0 / 512 / 1024
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py tests/test_gpu_cabi.py -x -q -k "not bench" > gpurun_out/r9_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r9_tests.log
grep -q "Memory access fault" gpurun_out/r9_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
ECOZ2_VQ_SPLIT_ACC_MAX_M=4096 timeout -k 10 600 python -m pytest tests/test_gpu_prefilter.py -x -q > gpurun_out/r9_tests2.log 2>&1; rc=$?
tail -4 gpurun_out/r9_tests2.log
[ $rc -ne 0 ] && exit $rc
for m in 0 512 4096 0 512 4096; do echo "== ECOZ2_VQ_SPLIT_ACC_MAX_M=$m"; ECOZ2_VQ_SPLIT_ACC_MAX_M=$m timeout -k 10 200 python tools/probe/ladder_real.py 2>&1 | tail -5; done
