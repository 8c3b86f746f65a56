#!/bin/bash
# round 4 profiles of the bench command: kernel trace + PMC passes for the M = 1024 / 512 / 256 levels
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
bash tools/profile_bench.sh r04 || exit 1
bash tools/profile_bench.sh r04m512 --codebook-size 512 || exit 1
bash tools/profile_bench.sh r04m256 --codebook-size 256 || exit 1
ls gpurun_out | grep r04 | head -40
