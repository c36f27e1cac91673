#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -x -q -m gpu -k "bn_backward or dgrad or upsample2x_and_backward or enc_skip or folded or first_layer" > gpurun_out/r5d_tests_new.txt 2>&1; tail -8 gpurun_out/r5d_tests_new.txt
timeout 900 python tools/ab_cfg.py base: none:fold_bn_bwd= d4a:fold_bn_bwd=e1b+d4a d3a:fold_bn_bwd=e1b+d3a d3b:fold_bn_bwd=e1b+d3b all:fold_bn_bwd=e1b+d4a+d3a+d3b > gpurun_out/r5d_ab.txt 2>&1; cat gpurun_out/r5d_ab.txt
timeout 600 python tools/archive/step_calls.py bf16x3 > gpurun_out/r5d_calls_x3.txt 2>&1; head -45 gpurun_out/r5d_calls_x3.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5d_tests_all.txt 2>&1; tail -5 gpurun_out/r5d_tests_all.txt
