#!/bin/bash
# tools/relink_attn.sh: recompile attention.hip only (production flags) and relink libsegclip_hip.so from the existing objects
# (build.sh rebuilds every translation unit when any header / .inc changed).
set -e
cd "$(dirname "$0")/../segclip_amd/csrc"
/opt/rocm/bin/hipcc $(cat build/.flags) -c attention.hip -o build/attention.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libsegclip_hip.so
echo "relinked"
