#!/bin/bash
# round 5: two blocks per turn in the fused two-stage sweep -- prefilter parity tests, then the bench line
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -x -q > gpurun_out/rl_tests.log 2>&1; rc=$?
tail -6 gpurun_out/rl_tests.log
grep -q "Memory access fault" gpurun_out/rl_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
ECOZ2_BENCH_SKIP_SMALL=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/rl_bench.json 2> gpurun_out/rl_bench.err || { tail -5 gpurun_out/rl_bench.err; exit 1; }
python tools/bench_digest.py "python bench.py --no-cpu-baseline"=gpurun_out/rl_bench.json | cut -c1-600
