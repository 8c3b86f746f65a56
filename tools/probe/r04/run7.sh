#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1000 python -m pytest tests/test_gpu_cabi.py -x -q -k "bench or bad_shard or in_process or rccl" > gpurun_out/r7_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r7_tests.log
grep -q "Memory access fault" gpurun_out/r7_tests.log && exit 1
exit $rc
