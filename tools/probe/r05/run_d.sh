#!/bin/bash
# kernel trace of the bench on the split pass: per-kernel durations at M = 1024 / 256
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_16M=1 ECOZ2_BENCH_SKIP_SMALL=1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 1024 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/rd_kt$m -- python3 $REPO/bench.py --no-cpu-baseline --no-extras --no-parity --steps 21 --codebook-size $m > $REPO/gpurun_out/rd_kt$m.json 2> $REPO/gpurun_out/rd_kt$m.err || { tail -5 $REPO/gpurun_out/rd_kt$m.err; exit 1; }
f=$(ls $REPO/gpurun_out/rd_kt$m/*/*kernel_stats.csv | head -1)
echo "== M = $m"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
