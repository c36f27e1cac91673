#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_scene.py -x -q -m gpu -k "two_chain or pickle or two_stream" > gpurun_out/r5b_tests_new.txt 2>&1; tail -5 gpurun_out/r5b_tests_new.txt
timeout 900 python tools/ab_cfg.py base: two_d2:fwd_chains=2,defer_product=2 two_d2_wg:fwd_chains=2,defer_product=2,fwd_chain2_role=wgrad two_d2_l2:fwd_chains=2,defer_product=2,fwd_chain_levels=2 two_d2_l4:fwd_chains=2,defer_product=2,fwd_chain_levels=4 two_d1:fwd_chains=2,defer_product=1 nohandoff:_diag_skip_handoff=1 noreduce:_diag_skip_reduce=1 > gpurun_out/r5b_ab.txt 2>&1; cat gpurun_out/r5b_ab.txt
cfg=fwd_chains=2,defer_product=2
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r5b_prof -o t --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 20 --windows 1 --engine-set $cfg > gpurun_out/r5b_rocprof.log 2>&1
python tools/timeline.py $(ls gpurun_out/r5b_prof/*kernel_trace.csv | head -1) 3 > gpurun_out/r5b_timeline_two_d2.txt 2>&1
rm -rf gpurun_out/r5b_prof
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5b_tests_all.txt 2>&1; tail -5 gpurun_out/r5b_tests_all.txt
timeout 900 python bench.py 2> gpurun_out/r5b_bench.err | grep '^{"metric"' | tail -1 > gpurun_out/r5b_bench.json; cut -c1-400 gpurun_out/r5b_bench.json; tail -3 gpurun_out/r5b_bench.err
