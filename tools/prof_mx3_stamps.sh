#!/bin/bash
# cycle stamps of csrc/conv_mx3.hip (-DMX3_PROF: workgroup 0 prints its waves' phase sums): builds e4s2024_amd/lib/libe4s_mx3prof.so here; on the GPU box:
#   E4S_HIP_LIB=e4s2024_amd/lib/libe4s_mx3prof.so python tools/probes/mx3_stamps.py
set -e
cd "$(dirname "$0")/.."
python -m e4s2024_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -fvisibility=hidden -DMX3_PROF \
    -c e4s2024_amd/csrc/conv_mx3.hip -o /tmp/conv_mx3_prof.o
objs=$(ls e4s2024_amd/build/*.o | grep -v conv_mx3.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o e4s2024_amd/lib/libe4s_mx3prof.so $objs /tmp/conv_mx3_prof.o
echo built e4s2024_amd/lib/libe4s_mx3prof.so
