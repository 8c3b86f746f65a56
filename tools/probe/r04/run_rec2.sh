#!/bin/bash
# the recorded accumulate: kernel trace of the ladder (records on), per-dispatch durations of the last ladder
R=$GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rec2_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/rec2.log 2>&1 || { tail -5 $R/gpurun_out/rec2.log; exit 1; }
cd $R
cat gpurun_out/rec2.log | tail -4
python3 tools/trace_gaps.py $(ls gpurun_out/rec2_kt/*/*_kernel_trace.csv | head -1) > gpurun_out/rec2_gaps.txt 2>&1
grep -n "k_pass_pre_lds\|k_reduce_records" gpurun_out/rec2_gaps.txt | tail -24
