#!/bin/bash
# round 5, final sources: more fuzz (other seeds) -- prefiltered mode 700 cases, general mode 300, HMM 100
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
{ timeout -k 10 700 python tools/fuzz_parity.py 700 77105 pre; } > gpurun_out/rq_fuzz_pre.txt 2>&1 || { tail -5 gpurun_out/rq_fuzz_pre.txt; exit 1; }
tail -1 gpurun_out/rq_fuzz_pre.txt
{ timeout -k 10 300 python tools/fuzz_parity.py 300 77106; } > gpurun_out/rq_fuzz_gen.txt 2>&1 || { tail -5 gpurun_out/rq_fuzz_gen.txt; exit 1; }
tail -1 gpurun_out/rq_fuzz_gen.txt
{ timeout -k 10 300 python tools/fuzz_parity.py 100 77107 hmm; } > gpurun_out/rq_fuzz_hmm.txt 2>&1 || { tail -5 gpurun_out/rq_fuzz_hmm.txt; exit 1; }
tail -1 gpurun_out/rq_fuzz_hmm.txt
