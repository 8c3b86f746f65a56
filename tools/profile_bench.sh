#!/bin/bash
# Profiles `bench.py` on the GPU box (run through gpurun from the repo root): one --kernel-trace/--stats run and
# separate PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ groups), each writing under gpurun_out/<tag>_*.
# tools/summarize_profiles.py turns the outputs into the summaries committed under profiles/.
#   usage: tools/profile_bench.sh <tag> [extra bench.py args]
set -e -o pipefail
TAG=${1:-r02}
shift || true
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ECOZ2_BENCH_SKIP_16M=1  # (the 16 M-frame extra is not part of any profile)
B="python3 $REPO/bench.py --no-cpu-baseline"
BP="$B --no-extras"  # counter passes: the timed region only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- $B --steps 21 --warmup 3 "$@" > "$OUT/${TAG}_kt.json" 2> "$OUT/${TAG}_kt.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_fetch" -- $BP --steps 6 --warmup 3 "$@" > "$OUT/${TAG}_fetch.json" 2> "$OUT/${TAG}_fetch.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_write" -- $BP --steps 6 --warmup 3 "$@" > "$OUT/${TAG}_write.json" 2> "$OUT/${TAG}_write.err"
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/${TAG}_sq1" -- $BP --steps 6 --warmup 3 "$@" > "$OUT/${TAG}_sq1.json" 2> "$OUT/${TAG}_sq1.err"
echo "sq1 done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d "$OUT/${TAG}_sq2" -- $BP --steps 6 --warmup 3 "$@" > "$OUT/${TAG}_sq2.json" 2> "$OUT/${TAG}_sq2.err"
echo "sq2 done"
