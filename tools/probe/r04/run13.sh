#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -x -q > gpurun_out/r13_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r13_tests.log
grep -q "Memory access fault" gpurun_out/r13_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for m in 0 1 0 1; do echo "== ECOZ2_VQ_RECORD_MOVERS=$m"; ECOZ2_VQ_RECORD_MOVERS=$m timeout -k 10 200 python tools/probe/ladder_real.py 2>&1 | tail -2; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r13_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/r13.log 2>&1
cd $R
f=$(ls -S gpurun_out/r13_kt/*/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f | grep -E "k_accum_ranges|k_pass_pre_lds|k_reduce_movers" | tail -18
