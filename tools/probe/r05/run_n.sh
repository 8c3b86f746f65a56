#!/bin/bash
# round 5: same-box A/B of the fused sorted pass: product / one block per turn / pipelined shuffle in the rows request
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_16M=1
for rep in 1 2; do
for v in product oneblk dmapipe; do
  lib=tools/probe/ab/$v/libecoz2vq.so; [ $v = product ] && lib=ecoz2rs_amd/csrc/libecoz2vq.so
  ECOZ2VQ_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/rn_$v.json 2> gpurun_out/rn_$v.err || { tail -5 gpurun_out/rn_$v.err; exit 1; }
  echo "== $v (run $rep)"; python tools/bench_digest.py "bench"=gpurun_out/rn_$v.json | grep -E "value|ladder" | cut -c1-520
done
done
