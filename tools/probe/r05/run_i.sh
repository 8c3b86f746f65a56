#!/bin/bash
# whole GPU suite + smoke + fuzz on the pruned / split tree
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/ri_tests.log 2>&1; rc=$?
tail -6 gpurun_out/ri_tests.log
grep -q "Memory access fault" gpurun_out/ri_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/ri_smoke.txt 2>&1 || { tail -5 gpurun_out/ri_smoke.txt; exit 1; }
tail -1 gpurun_out/ri_smoke.txt
{ timeout -k 10 400 python tools/fuzz_parity.py 300 9501 pre; } > gpurun_out/ri_fuzz.txt 2>&1 || { tail -5 gpurun_out/ri_fuzz.txt; exit 1; }
grep "done" gpurun_out/ri_fuzz.txt
