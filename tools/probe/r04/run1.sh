#!/bin/bash
# round 4, run 1: correctness of the new tile loop, then A/B against r03 and phase stamps
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -x -q > gpurun_out/r1_tests.log 2>&1 || { tail -30 gpurun_out/r1_tests.log; exit 1; }
tail -3 gpurun_out/r1_tests.log
bash tools/probe/ab/run_ab.sh base r03 base r03 > gpurun_out/r1_ab.log 2>&1
cat gpurun_out/r1_ab.log
ECOZ2VQ_LIB=$PWD/tools/probe/ab/stamp1/libecoz2vq.so timeout -k 10 300 python tools/probe/pre_stamps.py > gpurun_out/r1_stamps.log 2>&1
cat gpurun_out/r1_stamps.log
