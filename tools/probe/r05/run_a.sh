#!/bin/bash
# round 5, first GPU session: feasibility of skipping limb products + the small-corpus numbers, on the round-4 kernels
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 500 python tools/probe/skip_feasibility.py > gpurun_out/r05_skip_feasibility.txt 2>&1; echo "skip rc $?"
tail -30 gpurun_out/r05_skip_feasibility.txt
timeout -k 10 300 python tools/probe/small_corpus.py > gpurun_out/r05_small_corpus.txt 2>&1; echo "small rc $?"
tail -30 gpurun_out/r05_small_corpus.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_small_kt -- python3 $GRAFT_REPO_ROOT/tools/probe/small_corpus.py > $GRAFT_REPO_ROOT/gpurun_out/r05_small_kt.txt 2>&1; echo "kt rc $?"
