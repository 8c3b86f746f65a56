#!/bin/bash
# round 5 profiles of the bench command: kernel trace + PMC passes for the M = 1024 / 512 / 256 levels, and the plain sweep
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_SMALL=1
bash tools/profile_bench.sh r05 || exit 1
bash tools/profile_bench.sh r05np --no-prefilter || exit 1
bash tools/profile_bench.sh r05m512 --codebook-size 512 || exit 1
bash tools/profile_bench.sh r05m256 --codebook-size 256 || exit 1
ls gpurun_out | grep r05 | head -60
