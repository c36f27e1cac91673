#!/bin/bash
# run bench_conv conv for each variant lib; print totals + a few layers
for v in "$@"; do
  echo "== $v"; BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_$v.so python tools/bench_conv.py conv 2>&1 | grep -E "e1b fwd|e2b fwd|e3b dgrad|e4b dgrad|e5a fwd|d4a fwd|conv total"
done
