#!/bin/bash
# tools/build_exp_attn.sh: a second library segclip_amd/libsegclip_hip_exp.so whose attention.hip is compiled with
# -DSEGCLIP_EXPERIMENTS (timing ablations / s_memtime stamps: garbage results), every other object taken from the production build.
# The phase-stamp scripts under tools/debug load it by pointing segclip_amd._lib._LIB_PATH at it.
set -e
cd "$(dirname "$0")/../segclip_amd/csrc"
[ -f build/attention.o ] || ./build.sh
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSEGCLIP_EXPERIMENTS -c attention.hip -o /tmp/attention_exp.o
objs=$(ls build/*.o | grep -v "build/attention.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_exp.o -o ../libsegclip_hip_exp.so
echo "built $(realpath ../libsegclip_hip_exp.so)"
