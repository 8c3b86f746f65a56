#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 500 python tools/probe/skip_feasibility.py > gpurun_out/r05_skip_feasibility2.txt 2>&1; echo "skip rc $?"
grep "(b)\|level" gpurun_out/r05_skip_feasibility2.txt
