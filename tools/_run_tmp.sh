cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5e_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r5e_tests.txt 2>&1
timeout 1800 bash tools/make_profiles.sh r5_e > gpurun_out/r5e_make_profiles.log 2>&1
cp profiles/r5_e_* gpurun_out/ 2>/dev/null
cat gpurun_out/r5e_tests.txt; tail -3 gpurun_out/r5e_make_profiles.log | cut -c1-400
