#!/bin/bash
# run bench_conv conv for each variant lib; side-by-side per-layer microseconds
for v in "$@"; do
  BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_$v.so python tools/bench_conv.py conv > gpurun_out/ab_$v.txt 2>&1
done
python - "$@" <<'PY'
import sys,re
vs=sys.argv[1:]
tab={}
for v in vs:
    for line in open(f'gpurun_out/ab_{v}.txt'):
        m=re.match(r'(\S+ \S+)\s+N=.*?([\d.]+) us\s+([\d.]+) TF', line)
        if m: tab.setdefault(m.group(1),{})[v]=(float(m.group(2)),float(m.group(3)))
        m=re.match(r'conv total ([\d.]+) ms\s+([\d.]+)', line)
        if m: tab.setdefault('TOTAL(ms)',{})[v]=(float(m.group(1)),float(m.group(2)))
print(f'{"layer":12s}'+''.join(f'{v:>18s}' for v in vs))
for k,d in tab.items():
    print(f'{k:12s}'+''.join(f'{d[v][0]:10.1f}/{d[v][1]:6.0f} ' if v in d else ' '*18 for v in vs))
PY
