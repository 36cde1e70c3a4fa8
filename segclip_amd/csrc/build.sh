#!/bin/bash
# Build libsegclip_hip.so (gfx950 only) in-tree.  Usage: segclip_amd/csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libsegclip_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
mkdir -p build
pids=()
for f in gemm_f32.hip gemm_bf16.hip gemm_bf16_dma.hip layernorm.hip attention.hip misc.hip optim.hip; do
  $HIPCC $FLAGS -c $f -o build/${f%.hip}.o &
  pids+=($!)
done
$HIPCC $FLAGS -x hip -c capi.cpp -o build/capi.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
