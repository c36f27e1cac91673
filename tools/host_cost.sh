#!/bin/bash
# What an 8-rank launch costs the HOST (no multi-GPU hardware needed): host_enqueue_ms_per_step of bench.py's loop with
#   one rank, all cores  |  one rank pinned to ONE core  |  8 gloo ranks sharing this box's GPU, pinned to 8 cores (one per rank) | 8 ranks, all cores
# (the GPU is shared 8 ways in the last two, so their ms_per_step means nothing; 16 steps per window keep the enqueue inside the queue depth)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 16 --warmup 6 --windows 3 --no-extras --no-roofline --no-cpu-baseline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print({k: d.get(k) for k in ("n_gpus","ms_per_step","host_enqueue_ms_per_step","host_enqueue_ms_per_step_max_over_ranks")})'
echo "cores: $(nproc)"
echo "1 rank, all cores:";            python bench.py $A 2>/dev/null | python -c "$pick"
echo "1 rank, one core (taskset 0):"; taskset -c 0 python bench.py $A 2>/dev/null | python -c "$pick"
echo "8 gloo ranks on one GPU, cores 0-7:"; BENCH_ASSUME_DEVICES=8 BENCH_BACKEND=gloo taskset -c 0-7 python bench.py --gpus 8 $A 2>/dev/null | python -c "$pick"
echo "8 gloo ranks on one GPU, all cores:"; BENCH_ASSUME_DEVICES=8 BENCH_BACKEND=gloo python bench.py --gpus 8 $A 2>/dev/null | python -c "$pick"
python tools/probe_graph_fwd.py 2>&1 | grep -E "forward|step"
# 8 INDEPENDENT single-rank processes (no collective: gloo's all-reduce blocks the host, which is why the two gloo lines above read ~150 ms),
# one core each, sharing the GPU: the host loop of each under 8-process contention for the driver
echo "8 independent 1-rank processes, one core each, one shared GPU (host_enqueue_ms_per_step of each):"
for i in 0 1 2 3 4 5 6 7; do
  ( taskset -c $i python bench.py $A 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["host_enqueue_ms_per_step"],3), round(d["ms_per_step"],1))' ) &
done
wait
