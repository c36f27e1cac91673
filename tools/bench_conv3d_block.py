"""The conv3d leg of bench.py alone (BASELINE configs[3] shapes; tools/profile_legs.sh runs it under rocprofv3).
python tools/bench_conv3d_block.py [--iters 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
print(json.dumps(bench.conv3d_leg(torch.device('cuda', 0), iters=a.iters)))
