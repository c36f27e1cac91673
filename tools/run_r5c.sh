#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/archive/step_calls.py bf16x3 > gpurun_out/r5c_calls_x3.txt 2>&1; head -40 gpurun_out/r5c_calls_x3.txt
timeout 600 python tools/archive/step_calls.py bf16 > gpurun_out/r5c_calls_bf16.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r5c_tests_ddp.txt 2>&1; tail -5 gpurun_out/r5c_tests_ddp.txt
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_ddp.py > gpurun_out/r5c_tests_rest.txt 2>&1; tail -5 gpurun_out/r5c_tests_rest.txt
