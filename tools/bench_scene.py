"""Full-scene sliding-window inference throughput (BASELINE config 5 shape) on one MI355X.
usage: python tools/bench_scene.py [--size 4096] [--batch 64] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fabric_amd import BiDateNet                                     # noqa: E402
from fabric_amd.utils import inference as inf                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=4096)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--precision', default='bf16')
ap.add_argument('--one-lane', action='store_true', help='single-stream loop (clean per-kernel durations under rocprofv3)')
a = ap.parse_args()
torch.manual_seed(0)
model = BiDateNet(13, 2, precision=a.precision).cuda().eval()
h = w = a.size
t0 = time.time()
g = torch.Generator(device='cuda').manual_seed(3)
g1 = torch.randn(13, h, w, device='cuda', generator=g)
g2 = g1 + 0.3 * torch.randn(13, h, w, device='cuda', generator=g)
t1 = time.time()
torch.cuda.synchronize()
n = len(inf.tile_origins(h, w, 128)[0])
ts = False if a.one_lane else None
inf.predict_scene(model, g1, g2, 128, a.batch, two_streams=ts)
torch.cuda.synchronize()
best = 1e9
for _ in range(a.reps):
    t = time.time()
    m = inf.predict_scene(model, g1, g2, 128, a.batch, two_streams=ts)
    torch.cuda.synchronize()
    best = min(best, time.time() - t)
print(json.dumps({'workload': f'scene {h}x{w}x13 two dates, 128-px tiles, batch {a.batch}, {a.precision}',
                  'tiles': n, 'seconds': round(best, 4), 'tiles_per_s': round(n / best, 1),
                  'mpix_per_s': round(h * w / best / 1e6, 2), 'fwd_tflops': round(n * 23.14e9 / best / 1e12, 1),
                  'lanes': 1 if a.one_lane else 2, 'scene_gb': round(2 * 13 * h * w * 4 / 1e9, 2)}))
