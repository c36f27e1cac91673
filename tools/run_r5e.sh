#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -x -q -m gpu -k "bn_backward or folded" > gpurun_out/r5e_tests_new.txt 2>&1; tail -4 gpurun_out/r5e_tests_new.txt
timeout 900 python tools/ab_cfg.py base: e2b:fold_bn_bwd=e1b+d4a+e2b e1b:fold_bn_bwd=e1b > gpurun_out/r5e_ab.txt 2>&1; cat gpurun_out/r5e_ab.txt
bash tools/ab_lib.sh base bb3occ > gpurun_out/r5e_ablib.txt 2>&1; cat gpurun_out/r5e_ablib.txt
