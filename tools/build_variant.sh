#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DSEA_NO_REDO>"
# Builds global-motion-estimation_amd/csrc into tools/microbench/libgme_<name>.so (same C ABI as the tree's
# library) for same-box A/B runs with tools/ab.sh.
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
B=/tmp/gme_variant_$NAME; mkdir -p $B
SRC=global-motion-estimation_amd/csrc
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wno-unused-value $FLAGS"
pids=""
for f in gme_api gme_comm bbme_kernels bbme_fast bbme_sea bbme_sea_mse bbme_mfma bbme_walk16 gme_kernels synth_kernels; do
  $CXX -c $SRC/$f.hip -o $B/$f.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/libgme_$NAME.so $B/*.o -ldl
ls -la tools/microbench/libgme_$NAME.so
