#!/bin/bash
# Builds an A/B variant of libecoz2vq.so into tools/probe/ab/<name>/ (the product library is untouched):
#   tools/probe/ab/build_variant.sh <name> '<extra hipcc flags, e.g. -DE2VQ_PRE_STAMP=1>'
# Run with ECOZ2VQ_LIB=tools/probe/ab/<name>/libecoz2vq.so python bench.py ...
set -e
NAME=$1; FLAGS=$2
SRC=$(cd "$(dirname "$0")/../../../ecoz2rs_amd/csrc" && pwd)
OUT=$(cd "$(dirname "$0")" && pwd)/$NAME
mkdir -p "$OUT"
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950 $FLAGS"
for f in vq_device.hip vq_update.hip vq_pre_images.hip vq_prefilter.hip vq_sweep.hip hmm_device.hip vq_host.cpp vq_pass.cpp vq_group.cpp vq_entry.cpp vq_io.cpp seq_models.cpp hmm_host.cpp; do
  o=$OUT/$(basename ${f%.*}).o
  # only the kernel files see the flags' effect; the others are reused from the product build when present
  case $f in
    vq_prefilter.hip|vq_pre_images.hip|vq_device.hip|vq_update.hip|vq_sweep.hip) $CXX -x hip -c -o "$o" "$SRC/$f" & ;;
    *) if [ -f "$SRC/$(basename ${f%.*}).o" ]; then cp "$SRC/$(basename ${f%.*}).o" "$o"; else $CXX -x hip -c -o "$o" "$SRC/$f" & fi ;;
  esac
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libecoz2vq.so" "$OUT"/*.o -lpthread
echo "$OUT/libecoz2vq.so"
