#!/bin/bash
# k_cell_update A/B: tools/probe/ab/cu_ab.sh <variant>...  ("base" = product build); kernel traces land in gpurun_out/cu_ab_<v>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$R/tools/probe/ab/$v/libecoz2vq.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cu_ab_$v -- python3 $R/tools/probe/cu_rate.py > $R/gpurun_out/cu_ab_$v.log 2>&1 || exit 1
  grep "per round" $R/gpurun_out/cu_ab_$v.log
done
