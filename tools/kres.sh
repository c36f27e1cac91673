#!/bin/bash
# tools/kres.sh <file.hip> [name-filter] : one line per kernel of a csrc file -- VGPRs, scratch bytes, waves/SIMD (no GPU needed)
F=$1; K=${2:-.}
cd $(dirname $0)/../fabric_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $F -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('scr', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m: cur[key] = m.group(1)
    if 'lds' in cur:
        print('%4s vgpr %4s scratch %s waves  %s' % (cur.get('vgpr'), cur.get('scr'), cur.get('occ'), cur['name'].replace('unsigned short', 'bf16').replace('(ConvArgs)', '')))
        cur = {}
" | grep -E "$K"
rm -f /tmp/kres_$$.o
