# usage: ab_lib.sh libA libB  -> step time for each lib, alternating, same box
for rep in 1 2; do
for v in "$@"; do
  BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_$v.so python bench.py --steps 40 --warmup 10 --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
done; done
