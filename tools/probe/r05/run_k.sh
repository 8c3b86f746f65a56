#!/bin/bash
# round 5: timing experiments on the stamped sorted pass (tools/probe/sweep_exp.py)
cd $GRAFT_REPO_ROOT
EXP_MODES=${EXP_MODES:-0,32,64,96,0} ECOZ2VQ_LIB=tools/probe/ab/expA/libecoz2vq.so timeout -k 10 300 python tools/probe/sweep_exp.py > gpurun_out/rk_exp.txt 2>&1 || { tail -20 gpurun_out/rk_exp.txt; exit 1; }
cat gpurun_out/rk_exp.txt
