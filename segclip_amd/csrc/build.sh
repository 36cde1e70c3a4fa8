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
J=${BUILD_JOBS:-$(nproc)}
throttle() { while [ "$(jobs -rp | wc -l)" -ge "$J" ]; do wait -n || true; done; }
# the 8-phase GEMM (and likewise gemm_bf16_dma.hip, DMA_PART) is compiled once per operand layout (P8_PART 0..3), its dispatcher as part 4, and - only with
# -DSEGCLIP_P8_ABLATIONS among the flags - the main-loop ablation instances as part 5: the slow parts build in parallel
rm -f build/gemm_bf16.o
for part in 0 1 2 3 4; do   # the register-staged kernel first: its parts are the longest single jobs
  f=gemm_bf16.hip; o=build/gemm_bf16_part$part.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    throttle; $HIPCC $FLAGS -DGB_PART=$part -c $f -o $o &
    pids+=($!)
  fi
done
parts="0 1 2 3 4"; case "$FLAGS" in *SEGCLIP_P8_ABLATIONS*) parts="$parts 5";; esac
rm -f build/gemm_bf16_p8.o
for part in $parts; do
  f=gemm_bf16_p8.hip; o=build/gemm_bf16_p8_part$part.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    throttle; $HIPCC $FLAGS -DP8_PART=$part -c $f -o $o &
    pids+=($!)
  fi
done
for part in 0 1 2 3; do   # AGPR-accumulator 256x256 kernel (gemm_bf16_pq.hip): forward layout, data-gradient layout, dispatcher, weight-gradient layout
  f=gemm_bf16_pq.hip; o=build/gemm_bf16_pq_part$part.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    throttle; $HIPCC $FLAGS -DPQ_PART=$part -c $f -o $o &
    pids+=($!)
  fi
done
rm -f build/gemm_bf16_dma.o
for part in 0 1 2 3 4; do
  f=gemm_bf16_dma.hip; o=build/gemm_bf16_dma_part$part.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    throttle; $HIPCC $FLAGS -DDMA_PART=$part -c $f -o $o &
    pids+=($!)
  fi
done
for f in gemm_f32.hip layernorm.hip attention.hip misc.hip group_linear.hip head.hip center.hip optim.hip capi.cpp exec.cpp; do
  [ -f "$f" ] || continue
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    throttle; if [ "${f##*.}" = "cpp" ]; then $HIPCC $FLAGS -x hip -c $f -o $o & else $HIPCC $FLAGS -c $f -o $o & fi
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc -eq 0 ] || { echo "build failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
