#!/bin/bash
# round 5, first GPU call: new tests, two-chain forward A/B, timelines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -m gpu -k "two_chain or pickle" > gpurun_out/r5a_tests_new.txt 2>&1; tail -5 gpurun_out/r5a_tests_new.txt
timeout 600 python tools/ab_cfg.py base: two:fwd_chains=2 two_nodefer:fwd_chains=2,defer_product=0 two_wg:fwd_chains=2,fwd_chain2_role=wgrad two_l2:fwd_chains=2,fwd_chain_levels=2 two_l4:fwd_chains=2,fwd_chain_levels=4 > gpurun_out/r5a_ab.txt 2>&1; cat gpurun_out/r5a_ab.txt
for cfg in fwd_chains=1 fwd_chains=2; do
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r5a_prof_$cfg -o t --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 20 --windows 1 --engine-set $cfg > gpurun_out/r5a_rocprof_$cfg.log 2>&1
  python tools/timeline.py $(ls gpurun_out/r5a_prof_$cfg/*kernel_trace.csv | head -1) 3 > gpurun_out/r5a_timeline_$cfg.txt 2>&1
  rm -rf gpurun_out/r5a_prof_$cfg
  grep '^{"metric"' gpurun_out/r5a_rocprof_$cfg.log | cut -c1-300
done
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5a_tests_all.txt 2>&1; tail -5 gpurun_out/r5a_tests_all.txt
