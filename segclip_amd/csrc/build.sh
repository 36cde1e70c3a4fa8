#!/bin/bash
# Build libsegclip_hip.so (gfx950 only) in-tree.  Usage: segclip_amd/csrc/build.sh [extra hipcc flags]
# Incremental: a translation unit is recompiled only when it, a header, or the flags changed.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libsegclip_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
mkdir -p build
echo "$FLAGS" > build/.flags.new
if ! cmp -s build/.flags.new build/.flags 2>/dev/null; then rm -f build/*.o; mv build/.flags.new build/.flags; else rm -f build/.flags.new; fi
newest_hdr=$(ls -t *.h *.inc ../../include/*.h | head -1)
pids=()
for f in gemm_f32.hip gemm_bf16.hip gemm_bf16_dma.hip gemm_bf16_p8.hip layernorm.hip attention.hip misc.hip optim.hip capi.cpp; do
  [ -f "$f" ] || continue
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    if [ "${f##*.}" = "cpp" ]; then $HIPCC $FLAGS -x hip -c $f -o $o & else $HIPCC $FLAGS -c $f -o $o & fi
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
