#!/bin/bash
# Build libwfae.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result"
mkdir -p build
pids=()
for f in api gemm dconv norm_act ssim transformer wino forecast vit c1conv splitgemm c1b c1w g3b c1r c1rb; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ ../../include/wfae.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libwfae.so build/api.o build/gemm.o build/dconv.o build/norm_act.o build/ssim.o build/transformer.o build/wino.o build/forecast.o build/vit.o build/c1conv.o build/splitgemm.o build/c1b.o build/c1w.o build/g3b.o build/c1r.o build/c1rb.o
echo "built $(cd .. && pwd)/libwfae.so"
