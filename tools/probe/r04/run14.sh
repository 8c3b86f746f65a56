#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_cabi.py -x -q -k "other_prediction_orders or large_prediction or ragged or invalid" > gpurun_out/r14_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r14_tests.log
grep -q "Memory access fault" gpurun_out/r14_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/probe/generic_rate.py 2>&1 | tee gpurun_out/r14_orders.txt
timeout -k 10 600 python tools/fuzz_parity.py 150 4242 2>&1 | tail -3
