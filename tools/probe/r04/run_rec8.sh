#!/bin/bash
# incremental LDS-table passes (16 < M <= 128 on the plain sweep): the GPU suite, then the ladder with and without
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/rec8_tests.log 2>&1; rc=$?
tail -6 gpurun_out/rec8_tests.log
grep -q "Memory access fault" gpurun_out/rec8_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for r in 1 0 1 0; do
  echo "ECOZ2_VQ_PLAIN_INCREMENTAL=$r"
  ECOZ2_VQ_PLAIN_INCREMENTAL=$r timeout -k 10 300 python tools/probe/ladder_real.py || exit 1
done > gpurun_out/rec8_ladder.txt 2>&1
cat gpurun_out/rec8_ladder.txt
timeout -k 10 300 python tools/fuzz_parity.py 200 9201 > gpurun_out/rec8_fuzz.txt 2>&1 || { tail -5 gpurun_out/rec8_fuzz.txt; exit 1; }
tail -1 gpurun_out/rec8_fuzz.txt
