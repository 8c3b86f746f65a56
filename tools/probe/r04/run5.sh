#!/bin/bash
# full GPU suite + bench line
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_tests.log
grep -q "Memory access fault" gpurun_out/r5_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; tail -c 3000 gpurun_out/r5_bench.json
