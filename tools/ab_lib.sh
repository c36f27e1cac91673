#!/bin/bash
# Step time of two or more builds of the library on ONE box, alternating:  tools/ab_lib.sh [--precision bf16x3] name1 name2 ...
# (names of fabric_amd/csrc/variants/lib_<name>.so, e.g. built by tools/build_lib_variant.sh)
extra=""
if [ "$1" == "--precision" ]; then extra="--precision $2"; shift 2; fi
for rep in 1 2; do
for v in "$@"; do
  BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_$v.so python bench.py --steps 40 --warmup 10 --no-extras --no-roofline --no-cpu-baseline $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
done; done
