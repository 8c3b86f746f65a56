#!/bin/bash
# round 5, final sources: the profiles of the bench command (kernel trace + PMC passes for M = 1024 / 512 / 256 and the plain
# sweep), the phase stamps of the fused sorted pass (stamped library: tools/probe/ab/build_variant.sh stampF
# '-DE2VQ_SWEEP_STAMP=2'), the bench lines with the exchange timed, the scale check, one block per turn as the A/B
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
echo "== profiles"; ECOZ2_BENCH_SKIP_SMALL=1 bash tools/probe/r05/run_profiles.sh > gpurun_out/rf_profiles.log 2>&1 || { tail -5 gpurun_out/rf_profiles.log; exit 1; }
echo "== stamps"
ECOZ2VQ_LIB=tools/probe/ab/stampF/libecoz2vq.so timeout -k 10 300 python tools/probe/sweep_stamps.py > gpurun_out/rf_stamps.txt 2>&1 || { tail -5 gpurun_out/rf_stamps.txt; exit 1; }
echo "== A/B one block per turn"
for v in product oneblk; do
  lib=tools/probe/ab/$v/libecoz2vq.so; [ $v = product ] && lib=ecoz2rs_amd/csrc/libecoz2vq.so
  ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_16M=1 ECOZ2VQ_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/rf_ab_$v.json 2> gpurun_out/rf_ab_$v.err || { tail -5 gpurun_out/rf_ab_$v.err; exit 1; }
done
echo "== bench lines, scale check"
bash tools/probe/r05/run_j.sh > gpurun_out/rf_runj.log 2>&1 || { tail -8 gpurun_out/rf_runj.log; exit 1; }
tail -30 gpurun_out/rf_runj.log | cut -c1-300
