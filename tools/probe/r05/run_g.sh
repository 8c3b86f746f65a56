#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
for v in "$@"; do
ECOZ2VQ_LIB=$PWD/tools/probe/ab/$v/libecoz2vq.so timeout -k 10 300 python tools/probe/sweep_stamps.py > gpurun_out/rg_$v.txt 2>&1; echo "$v rc $?"; grep "pass [23]" gpurun_out/rg_$v.txt | cut -c1-420
done
