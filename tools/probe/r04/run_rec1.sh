#!/bin/bash
# the recorded accumulate: parity first, then the ladder with and without it, then a kernel trace of the ladder
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/rec1_tests.log 2>&1; rc=$?
tail -15 gpurun_out/rec1_tests.log
grep -q "Memory access fault" gpurun_out/rec1_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for r in 1 0; do
  echo "ECOZ2_VQ_RECORDS=$r"
  ECOZ2_VQ_RECORDS=$r timeout -k 10 300 python tools/probe/ladder_real.py || exit 1
done > gpurun_out/rec1_ladder.txt 2>&1
cat gpurun_out/rec1_ladder.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rec1_kt -o run -- python3 $GRAFT_REPO_ROOT/tools/probe/ladder_real.py > $GRAFT_REPO_ROOT/gpurun_out/rec1_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/rec1_kt/*/*_kernel_trace.csv | head -1) > gpurun_out/rec1_gaps.txt 2>&1; tail -3 gpurun_out/rec1_gaps.txt
ls gpurun_out/rec1_kt/ | head
