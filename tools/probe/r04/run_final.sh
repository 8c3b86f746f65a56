#!/bin/bash
# round 4, final: the whole GPU suite, the fuzz modes, the bench line and its collective variants
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/rf_tests.log 2>&1; rc=$?
tail -4 gpurun_out/rf_tests.log
grep -q "Memory access fault" gpurun_out/rf_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
{ timeout -k 10 300 python tools/fuzz_parity.py 400 9001; timeout -k 10 400 python tools/fuzz_parity.py 300 9002 pre; timeout -k 10 200 python tools/fuzz_parity.py 100 9003 hmm; } > gpurun_out/rf_fuzz.txt 2>&1
tail -3 gpurun_out/rf_fuzz.txt
python bench.py > gpurun_out/rf_bench.json 2> gpurun_out/rf_bench.err || { tail -5 gpurun_out/rf_bench.err; exit 1; }
python bench.py --force-collective --no-cpu-baseline > gpurun_out/rf_bench_fc.json 2> gpurun_out/rf_bench_fc.err
python bench.py --gpus 2 --in-process --no-cpu-baseline > gpurun_out/rf_bench_inproc2.json 2> gpurun_out/rf_bench_inproc2.err
python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-extras > gpurun_out/rf_bench_gloo2.json 2> gpurun_out/rf_bench_gloo2.err
python tools/probe/scale_check.py > gpurun_out/rf_scale_check.txt 2>&1
tail -12 gpurun_out/rf_scale_check.txt
ls -la gpurun_out/rf_*
