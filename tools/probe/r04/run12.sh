#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
R=$GRAFT_REPO_ROOT
export ECOZ2_VQ_SPLIT_ACC_MAX_M=4096
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r12_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/r12.log 2>&1
cd $R
tail -3 gpurun_out/r12.log
f=$(ls -S gpurun_out/r12_kt/*/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f | grep -E "k_accum_ranges|k_pass_pre_lds" | tail -18
