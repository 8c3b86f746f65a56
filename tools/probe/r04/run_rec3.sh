#!/bin/bash
# the recorded accumulate: parity (prefilter tests), ladder A/B, kernel trace of the ladder
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu > gpurun_out/rec3_tests.log 2>&1; rc=$?
tail -5 gpurun_out/rec3_tests.log
grep -q "Memory access fault" gpurun_out/rec3_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for r in 1 0; do
  echo "ECOZ2_VQ_RECORDS=$r"
  ECOZ2_VQ_RECORDS=$r timeout -k 10 300 python tools/probe/ladder_real.py || exit 1
done > gpurun_out/rec3_ladder.txt 2>&1
cat gpurun_out/rec3_ladder.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rec3_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/rec3.log 2>&1 || { tail -5 $R/gpurun_out/rec3.log; exit 1; }
cd $R
python3 tools/trace_gaps.py $(ls gpurun_out/rec3_kt/*/*_kernel_trace.csv | head -1) > gpurun_out/rec3_gaps.txt 2>&1
grep -n "k_pass_pre_lds\|k_reduce_records" gpurun_out/rec3_gaps.txt | tail -9
