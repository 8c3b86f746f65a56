#!/bin/bash
# the recorded accumulate: reducing workgroups, seeded first passes from M = 256 / 128, the prefilter from M = 64 / 128
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
run() { echo "$*"; env "$@" timeout -k 10 300 python tools/probe/ladder_real.py | tail -2 || exit 1; }
{
run ECOZ2_VQ_REC_WG=512
run ECOZ2_VQ_REC_WG=768
run ECOZ2_VQ_REC_WG=1024
run ECOZ2_VQ_REC_WG=1536
run ECOZ2_VQ_FAMILY_MIN_M=256
run ECOZ2_VQ_FAMILY_MIN_M=128 ECOZ2_VQ_PREFILTER_MIN_M=128
run ECOZ2_VQ_FAMILY_MIN_M=64 ECOZ2_VQ_PREFILTER_MIN_M=64
run ECOZ2_VQ_FAMILY_MIN_M=4096 ECOZ2_VQ_PREFILTER_MIN_M=128
} > gpurun_out/rec6_ladder.txt 2>&1
cat gpurun_out/rec6_ladder.txt
