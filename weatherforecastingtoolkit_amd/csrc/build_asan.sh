#!/bin/bash
# Host-only AddressSanitizer + UBSan build of the C-ABI layer (argument validation, workspace arithmetic, launch
# geometry code that runs on the CPU before any kernel is launched): build_asan/libwfae_asan.so.  Only the HOST side is
# instrumented (-fno-gpu-sanitize; device code is the plain -O1 build) — it exists for tests/test_abi_cpu.py on
# the CPU box only (GPU sanitizers are not available on the pool).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -fno-gpu-sanitize -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wno-unused-result"
mkdir -p build_asan
pids=()
for f in api gemm dconv norm_act ssim transformer wino forecast vit c1conv splitgemm c1b c1w g3b c1r c1rb; do
  if [ ! -f build_asan/$f.o ] || [ $f.hip -nt build_asan/$f.o ] || [ common.h -nt build_asan/$f.o ] || [ ../../include/wfae.h -nt build_asan/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build_asan/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -fno-gpu-sanitize -shared -fPIC -fsanitize=address,undefined -shared-libasan -o build_asan/libwfae_asan.so build_asan/*.o
echo "built $(pwd)/build_asan/libwfae_asan.so"
