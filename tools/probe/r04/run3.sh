#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
ECOZ2VQ_LIB=$PWD/tools/probe/ab/stamp1/libecoz2vq.so timeout -k 10 300 python tools/probe/pre_stamps.py > gpurun_out/r3_stamps.log 2>&1
cat gpurun_out/r3_stamps.log
grep -q "Memory access fault" gpurun_out/r3_stamps.log && exit 1
exit 0
