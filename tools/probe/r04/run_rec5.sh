#!/bin/bash
# per-pass fixed cost: the event behind the statistics kernel (A/B), kernel trace of the ladder without it
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
for ev in 0 1 0 1; do
  echo "ECOZ2_VQ_STATS_EVENT=$ev"
  ECOZ2_VQ_STATS_EVENT=$ev timeout -k 10 300 python tools/probe/ladder_real.py || exit 1
done > gpurun_out/rec5_ladder.txt 2>&1
cat gpurun_out/rec5_ladder.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rec5_kt -- python3 $R/tools/probe/ladder_real.py > $R/gpurun_out/rec5.log 2>&1 || { tail -5 $R/gpurun_out/rec5.log; exit 1; }
cd $R
python3 tools/trace_gaps.py $(ls gpurun_out/rec5_kt/*/*_kernel_trace.csv | head -1) > gpurun_out/rec5_gaps.txt 2>&1
tail -30 gpurun_out/rec5_gaps.txt
