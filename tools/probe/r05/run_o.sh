#!/bin/bash
# round 5: the M = 128 level on the fused sorted pass (ECOZ2_VQ_ACCUMULATE=sorted) against the default (round 4's kernel there)
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_16M=1
for rep in 1 2; do
for acc in auto sorted; do
  ECOZ2_VQ_ACCUMULATE=$acc timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ro_$acc.json 2> gpurun_out/ro_$acc.err || { tail -5 gpurun_out/ro_$acc.err; exit 1; }
  echo "== $acc (run $rep)"; python tools/bench_digest.py "bench"=gpurun_out/ro_$acc.json | grep -E "value|ladder" | cut -c1-520
done
done
