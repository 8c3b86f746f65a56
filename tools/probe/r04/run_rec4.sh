#!/bin/bash
# the recorded accumulate: number of reducing workgroups (incremental passes / full passes)
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
for wg in 256 512 1024 2048 4096; do
  echo "ECOZ2_VQ_REC_WG_FEW=$wg ECOZ2_VQ_REC_WG_MANY=$wg"
  ECOZ2_VQ_REC_WG_FEW=$wg ECOZ2_VQ_REC_WG_MANY=$wg timeout -k 10 300 python tools/probe/ladder_real.py | tail -2 || exit 1
done > gpurun_out/rec4_ladder.txt 2>&1
cat gpurun_out/rec4_ladder.txt
