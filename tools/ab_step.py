"""A/B of engine switches inside ONE process (boxes differ by +-10 %, so variants must share a run):
alternates blocks of training steps with each setting and prints the median ms/step per setting.
usage: python tools/ab_step.py attr=val0,val1 [blocks] [steps_per_block]      e.g. wgrad_blocks=0,192"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep

attr, vals = sys.argv[1].split('=')
vals = [int(v) for v in vals.split(',')]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spb = int(sys.argv[3]) if len(sys.argv) > 3 else 15
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
eng = model.engine()
for _ in range(10): step.step(x1, x2, lbl)
torch.cuda.synchronize()
res = {v: [] for v in vals}
for b in range(blocks):
    for v in vals:
        setattr(eng, attr, bool(v) if isinstance(getattr(eng, attr), bool) else v)
        for _ in range(3): step.step(x1, x2, lbl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(spb): step.step(x1, x2, lbl)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / spb)
for v in vals:
    print(f'{attr}={v}: median {statistics.median(res[v]):.3f} ms/step  (min {min(res[v]):.3f}, max {max(res[v]):.3f})  -> {B / statistics.median(res[v]) * 1e3:.0f} pairs/s')
