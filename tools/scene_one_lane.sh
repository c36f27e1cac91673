cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/sl_prof -o p --output-format csv -- python tools/bench_scene.py --size 6144 --batch 256 --reps 2 --one-lane > gpurun_out/sl.log 2>&1
python tools/timeline.py $(ls gpurun_out/sl_prof/*kernel_trace.csv | head -1) 2 gather_tiles_kernel > gpurun_out/r6_a_scene_one_lane_timeline.txt 2>&1
grep '^{' gpurun_out/sl.log
rm -rf gpurun_out/sl_prof
