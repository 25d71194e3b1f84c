#!/bin/bash
# Tuning builds of the DMA-fed masked / plain-convolution kernel (csrc/modconv_mx.hip, -DMX_ABL=bits, see its header): one library per ablation in
# e4s2024_amd/lib/libe4s_abl<bits>.so (git-ignored, travels to the GPU box), selected with E4S_HIP_LIB.  Results of such a library are meaningless — only its
# kernel time is: what is left when a part of the loop is taken out tells which part bounds it.
#   tools/build_abl.sh 1 8 9 ...;   E4S_HIP_LIB=e4s2024_amd/lib/libe4s_abl1.so python tools/time_enc_mx.py
set -e
cd "$(dirname "$0")/.."
python -m e4s2024_amd.build > /dev/null
build_one() { a=$1
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -fvisibility=hidden -DMX_ABL=$a \
      -c e4s2024_amd/csrc/modconv_mx.hip -o /tmp/modconv_mx_abl$a.o
  objs=$(ls e4s2024_amd/build/*.o | grep -v modconv_mx.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o e4s2024_amd/lib/libe4s_abl$a.so $objs /tmp/modconv_mx_abl$a.o
  echo "built e4s2024_amd/lib/libe4s_abl$a.so"; }
for a in "$@"; do build_one $a & if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi; done; wait
