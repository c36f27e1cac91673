#!/bin/bash
# HBM traffic of every kernel of one run: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots),
# kernel-trace/stats off (gpurun refuses --pmc with trace domains).
#   tools/pmc_traffic.sh                               the bf16 training loop  -> gpurun_out/pmc_traffic.json
#   tools/pmc_traffic.sh <out.json> <units> <cmd...>   any other leg; <units> = steps / batches the command runs (recorded in _meta)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_traffic.json}
UNITS=${2:-5}
if [ $# -ge 3 ]; then shift 2; CMD="$*"; else CMD="python bench.py --steps 3 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-extras"; fi
rm -rf gpurun_out/pmct
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmct/f -o p --output-format csv -- $CMD > gpurun_out/pmct_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmct/w -o p --output-format csv -- $CMD > gpurun_out/pmct_w.log 2>&1
PMC_OUT="$OUT" PMC_UNITS="$UNITS" PMC_CMD="$CMD" python - <<'PY'
import csv, glob, json, collections, os
out = collections.defaultdict(lambda: {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
for kind in ('f', 'w'):
    for f in glob.glob(f'gpurun_out/pmct/{kind}/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            out[k][r['Counter_Name']] += float(r['Counter_Value'])
            if kind == 'f': out[k]['launches'] += 1
res = {}
for k, v in out.items():
    n = max(1, v['launches'])
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2
    res[k] = {'launches': n, 'fetch_bytes_per_launch_raw': v['FETCH_SIZE'] * 1024 / n,
              'fetch_bytes_per_launch_corrected': 2 * v['FETCH_SIZE'] * 1024 / n,
              'write_bytes_per_launch': v['WRITE_SIZE'] * 1024 / n}
res['_meta'] = {'steps': int(os.environ['PMC_UNITS']), 'command': 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- ' + os.environ['PMC_CMD']}
json.dump(res, open(os.environ['PMC_OUT'], 'w'), indent=1)
for k in sorted([k for k in res if k != '_meta'], key=lambda k: -res[k]['fetch_bytes_per_launch_corrected'] * res[k]['launches'])[:8]:
    print(k[:60], {a: round(b / 1e6, 1) if a != 'launches' else b for a, b in res[k].items()})
PY
