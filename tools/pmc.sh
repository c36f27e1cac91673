#!/bin/bash
# usage: tools/pmc.sh <outdir> <python args...>   -- collects SQ counters in separate passes (no trace domains with --pmc)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC"; do
  rocprofv3 --pmc $set -d $out/p$i -o pmc --output-format csv -- python "$@" > $out.log$i 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('$out/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:70]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'conv3x3' in k or 'wgrad_kernel' in k:
        print(k)
        for c,val in sorted(v.items()): print(f'   {c:28s} {val:.4g}')
PY
