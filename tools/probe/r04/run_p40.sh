#!/bin/bash
# P = 40 on the LDS-staged pass (seven waves) + recorded accumulate: tests, then the rate against the round-2 kernel
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu -k "order" > gpurun_out/p40_tests.log 2>&1; rc=$?
tail -5 gpurun_out/p40_tests.log
grep -q "Memory access fault" gpurun_out/p40_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for r in 1 0; do echo "ECOZ2_VQ_RECORDS=$r"; ECOZ2_VQ_RECORDS=$r timeout -k 10 300 python tools/probe/orders_rate.py 2>&1 | tail -3; done | tee gpurun_out/p40_rate.txt
