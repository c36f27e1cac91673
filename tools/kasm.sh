#!/bin/bash
# tools/kasm.sh <file.hip> <mangled-kernel-substring> : compile one csrc file for gfx950 with -save-temps, print the kernel's
# resource usage and leave its ISA in /tmp/kasm/<substring>.s  (no GPU needed)
set -e
F=$1; K=$2
mkdir -p /tmp/kasm && cd /root/repo/fabric_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-shift-negative-value \
  -c $F -o /tmp/kasm/${F%.hip}.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2> /tmp/kasm/log.txt || { grep -v "^remark" /tmp/kasm/log.txt | head -40; exit 1; }
grep -v "^remark" /tmp/kasm/log.txt | head -20
S=/tmp/kasm/${F%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
grep -A9 "Function Name: .*$K" /tmp/kasm/log.txt | grep -i "name\|SGPRs:\|VGPRs:\|AGPRs\|Scratch\|Occupancy\|Spill" | sed 's/ \[-Rpass.*//; s/remark: [^ ]* *//'
awk -v k="$K" '$0 ~ "^_Z[^ ]*" k "[^ ]*:" {p=1} p {print} p && /s_endpgm/ {exit}' $S > /tmp/kasm/$K.s
wc -l /tmp/kasm/$K.s
