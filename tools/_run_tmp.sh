#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_train.py tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r5m_tests.txt 2>&1; tail -3 gpurun_out/r5m_tests.txt
timeout 600 python bench.py --no-extras --no-cpu-baseline 2> gpurun_out/r5m_bench.err | grep '^{"metric"' | tail -1 > gpurun_out/r5m_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r5m_bench.json')); r=d['roofline']; print(d['value'], r['kernel'], r['frac'], r['launches_per_step'], r['avg_launch_us'], r['rocprof_avg_launch_us'])
for k,v in r['families'].items(): print(k, v['launches_per_step'], v['ms_per_step'], round(v['TFLOPs']), v['rocprof_avg_launch_us'])"
