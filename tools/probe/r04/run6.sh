#!/bin/bash
# three waves per SIMD for the LDS-table levels: parity, then the ladder both ways
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 600 python -m pytest tests/test_gpu_prefilter.py -x -q -k "small_codebook" > gpurun_out/r6_tests.log 2>&1 || { tail -20 gpurun_out/r6_tests.log; exit 1; }
tail -2 gpurun_out/r6_tests.log
for w in 1 0 1 0; do echo "== ECOZ2_VQ_WAVES3=$w"; ECOZ2_VQ_WAVES3=$w timeout -k 10 200 python tools/probe/ladder.py 2>&1 | grep -E "M= *(16|32|64|128|256) " ; done
