#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/make_profiles.sh r5_a
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5f_tests_all.txt 2>&1; tail -3 gpurun_out/r5f_tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5f_smoke.txt 2>&1; tail -3 gpurun_out/r5f_smoke.txt
timeout 600 python tools/ab_skip.py > gpurun_out/r5f_ab_skip.txt 2>&1; cat gpurun_out/r5f_ab_skip.txt
