"""Does running the dz chain on a HIGH-priority stream (weight gradients stay on a normal-priority side stream) help?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
print('priority range', lo, hi)
streams = {'default': None, 'high': torch.cuda.Stream(priority=-1)}
res = {k: [] for k in streams}
for rep in range(4):
    for name, s in streams.items():
        ctx = torch.cuda.stream(s) if s is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            for _ in range(4): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(15): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 15)
for k, v in res.items(): print(k, f'median {statistics.median(v):.3f} ms/step', [round(x, 3) for x in v])
