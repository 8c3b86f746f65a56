#!/bin/bash
# round 5: two blocks per turn -- stamps of the new kernel, experiments, the bench line
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
ECOZ2VQ_LIB=tools/probe/ab/expA/libecoz2vq.so timeout -k 10 300 python tools/probe/sweep_stamps.py > gpurun_out/rm_stamps.txt 2>&1 || { tail -5 gpurun_out/rm_stamps.txt; exit 1; }
grep -v "pass 1" gpurun_out/rm_stamps.txt | cut -c1-420
EXP_MODES=0,3,6 ECOZ2VQ_LIB=tools/probe/ab/expA/libecoz2vq.so timeout -k 10 300 python tools/probe/sweep_exp.py > gpurun_out/rm_exp.txt 2>&1 || { tail -5 gpurun_out/rm_exp.txt; exit 1; }
cut -c1-330 gpurun_out/rm_exp.txt
ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_16M=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/rm_bench.json 2> gpurun_out/rm_bench.err || { tail -5 gpurun_out/rm_bench.err; exit 1; }
python tools/bench_digest.py "python bench.py --no-cpu-baseline"=gpurun_out/rm_bench.json | cut -c1-600
