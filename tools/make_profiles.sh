#!/bin/bash
# Round profile artefacts, all from ONE box (run through gpurun):  tools/make_profiles.sh <tag>   e.g. r1_d (the tag bench.py names in _rocprof_avg)
#   gpurun_out/<tag>_bench.json                 default bench.py line (roofline + cpu_baseline)
#   gpurun_out/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_bench_under_rocprof.json   the bench line printed under the profiler
#   gpurun_out/pmc_traffic.json                 FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes)
tag=${1:-r1_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json profiles/r1_pmc_traffic.json      # bench.py reads the traffic figures from here
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o ${tag} --output-format csv -- python bench.py > gpurun_out/${tag}_rocprof.log 2>&1
grep '^{"metric"' gpurun_out/${tag}_rocprof.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
cp gpurun_out/${tag}_prof/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv   # bench.py quotes this file's average beside its live event figure
python bench.py 2> gpurun_out/${tag}_bench.err | grep '^{"metric"' | tail -1 > gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_bench.json | cut -c1-400
head -12 gpurun_out/${tag}_kernel_stats.csv
