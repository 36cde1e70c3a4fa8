#!/bin/bash
# tools/asm_check.sh <file.hip> [extra hipcc flags, e.g. -DP8_PART=1] : compile one translation unit for gfx950 with -save-temps into /tmp/asm and print
# the per-kernel register / spill / scratch summary.
set -e
f=$(realpath "$1"); b=$(basename "$f" .hip); shift
extra="$*"; case "$b" in gemm_bf16_p8) case "$extra" in *P8_PART*) ;; *) extra="$extra -DP8_PART=0";; esac;;
  gemm_bf16_pq) case "$extra" in *PQ_PART*) ;; *) extra="$extra -DPQ_PART=0";; esac;;
  gemm_bf16) case "$extra" in *GB_PART*) ;; *) extra="$extra -DGB_PART=0";; esac;;
  gemm_bf16_dma) case "$extra" in *DMA_PART*) ;; *) extra="$extra -DDMA_PART=0";; esac;; esac
mkdir -p /tmp/asm/$b && cd /tmp/asm/$b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra -save-temps -c "$f" -o $b.o 2>&1 | grep -v "^$" | grep -v "loop not unrolled\|warnings\? generated" || true
S=$b-hip-amdgcn-amd-amdhsa-gfx950.s
grep "^    \.name:\|\.vgpr_count\|vgpr_spill\|private_segment_fixed_size:\|\.sgpr_count" $S | paste - - - - - | sed 's/  */ /g'
echo "asm: /tmp/asm/$b/$S"
