#!/bin/bash
# Round profile artefacts, all from ONE box (run through gpurun):  tools/make_profiles.sh <tag>   e.g. r2_a
#   profiles/<tag>_pmc_traffic.json            FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, tools/pmc_traffic.sh)
#   profiles/<tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of  python bench.py --no-extras --no-cpu-baseline
#                                              (the timed training loop only: the extra legs reuse the same kernel names)
#   profiles/<tag>_bench_under_rocprof.json    the bench line printed under the profiler
#   profiles/<tag>_bench.json                  default bench.py line (roofline + cpu_baseline + extra legs), un-profiled
# bench.py reads the newest r*_pmc_traffic.json / r*_kernel_stats.csv, so the traffic passes run first.
tag=${1:-r2_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o ${tag} --output-format csv -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/${tag}_rocprof.log 2>&1
grep '^{"metric"' gpurun_out/${tag}_rocprof.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
cp gpurun_out/${tag}_prof/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv
python tools/timeline.py $(ls gpurun_out/${tag}_prof/*kernel_trace.csv | head -1) 3 > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/${tag}_prof gpurun_out/pmct
python bench.py 2> gpurun_out/${tag}_bench.err | grep '^{"metric"' | tail -1 > gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_bench.json | cut -c1-600
head -12 gpurun_out/${tag}_kernel_stats.csv
