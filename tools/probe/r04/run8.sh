#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1000 python -m pytest tests/test_gpu_hmm.py -x -q > gpurun_out/r8_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r8_tests.log
grep -q "Memory access fault" gpurun_out/r8_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/probe/hmm_rate.py > gpurun_out/r8_hmm_rate.txt 2>&1
cat gpurun_out/r8_hmm_rate.txt
