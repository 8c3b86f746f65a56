#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_VQ_SPLIT_ACC_MAX_M=1024
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r10_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/r10.log 2>&1
cd $R
f=$(ls -S gpurun_out/r10_kt/*/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f | grep -E "k_accum_ranges|k_pass_pre_lds|k_pass_mfma|k_cell_update|k_pass_prologue|k_pre_codebook" | tail -60
