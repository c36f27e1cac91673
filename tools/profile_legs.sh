#!/bin/bash
# rocprof evidence for the legs that are NOT the bf16 training loop (run through gpurun):  tools/profile_legs.sh <tag> [legs]
#   legs (default: all)   x3      python bench.py --precision bf16x3 (the setting inside north_star's 1e-3 logit bar)
#                         scene   tools/bench_scene.py (BASELINE configs[4]; kernel stats at 10000^2 on ONE lane -- the product runs two, whose kernels
#                                 overlap and would double every duration --, PMC at 4096^2: same 256-tile launches)
#                         conv3d  tools/bench_conv3d_block.py (BASELINE configs[3] shapes)
#   profiles/<tag>_<leg>_kstats.csv           rocprofv3 --kernel-trace --stats
#   profiles/<tag>_<leg>_pmc.json             FETCH_SIZE (x2) / WRITE_SIZE per kernel, separate --pmc passes (tools/pmc_traffic.sh)
#   profiles/<tag>_<leg>_line.json            what the leg printed under the kernel-trace pass
tag=${1:-r6_x}; shift
legs=${*:-x3 scene conv3d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for leg in $legs; do
    case $leg in
        x3)     cmd="python bench.py --precision bf16x3 --steps 10 --warmup 4 --windows 1 --no-cpu-baseline --no-extras"
                pmc="python bench.py --precision bf16x3 --steps 3 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-extras"; units=5; marker=pack_input ;;
        scene)  cmd="python tools/bench_scene.py --size 10000 --batch 256 --reps 2 --one-lane"      # one lane: per-kernel durations without the other lane's kernels beside them
                pmc="python tools/bench_scene.py --size 4096 --batch 256 --reps 1"; units=8; marker=gather_tiles_kernel ;;
        conv3d) cmd="python tools/bench_conv3d_block.py"
                pmc="python tools/bench_conv3d_block.py --iters 1"; units=2; marker=pack_weights ;;
    esac
    rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_${leg}_prof -o p --output-format csv -- $cmd > gpurun_out/${tag}_${leg}_rocprof.log 2>&1
    grep '^{' gpurun_out/${tag}_${leg}_rocprof.log | tail -1 > profiles/${tag}_${leg}_line.json
    cp gpurun_out/${tag}_${leg}_prof/*kernel_stats.csv profiles/${tag}_${leg}_kstats.csv
    python tools/timeline.py $(ls gpurun_out/${tag}_${leg}_prof/*kernel_trace.csv | head -1) 1 $marker > gpurun_out/${tag}_${leg}_timeline.txt 2>&1
    rm -rf gpurun_out/${tag}_${leg}_prof
    bash tools/pmc_traffic.sh profiles/${tag}_${leg}_pmc.json $units $pmc > gpurun_out/${tag}_${leg}_pmc.txt 2>&1
    rm -rf gpurun_out/pmct
    cp profiles/${tag}_${leg}_* gpurun_out/
    echo "== $leg"; cut -c1-400 profiles/${tag}_${leg}_line.json; head -8 profiles/${tag}_${leg}_kstats.csv | cut -c1-200
done
