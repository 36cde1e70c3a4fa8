#!/bin/bash
# tools/relink_attn_variant.sh <name>: like tools/relink_attn.sh, into segclip_amd/libsegclip_hip_<name>.so (for tools/debug/attn_ab.py)
set -e
cd "$(dirname "$0")/../segclip_amd/csrc"
/opt/rocm/bin/hipcc $(cat build/.flags) -c attention.hip -o /tmp/attention_$1.o
objs=$(ls build/*.o | grep -v "build/attention.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_$1.o -o ../libsegclip_hip_$1.so
echo "built libsegclip_hip_$1.so"
