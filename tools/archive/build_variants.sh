#!/bin/bash
# build conv3x3 variants: tools/archive/build_variants.sh name "-DF_X=0 ..." [name flags]...
cd /root/repo/fabric_amd/csrc && mkdir -p variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -Rpass-analysis=kernel-resource-usage -c conv3x3.hip -o variants/conv_$name.o 2>variants/conv_$name.log; grep -E "error" variants/conv_$name.log
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/conv_$name.o wgrad.o bn.o fuse.o head.o scene.o x3.o -o variants/lib_$name.so && echo built $name &
done
wait
