#!/bin/bash
# Build a whole-library variant with extra compiler flags:  tools/build_lib_variant.sh name "-DFOO=1" [name flags]...
# -> fabric_amd/csrc/variants/lib_<name>.so (git-ignored; travels with gpurun; select with BIDATE_LIB or tools/ab_lib.sh)
cd /root/repo/fabric_amd/csrc && mkdir -p variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  (
    objs=""
    for s in conv3x3 wgrad bn fuse head scene x3; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $s.hip -o variants/${s}_$name.o 2>variants/${s}_$name.log || { echo "FAILED $s ($name)"; grep -m5 error variants/${s}_$name.log; }
      objs="$objs variants/${s}_$name.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/lib_$name.so && echo built $name
  ) &
done
wait
