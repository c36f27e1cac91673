"""Stream priorities of the two queues of a training step (chain = forward / loss / dz chain / SGD, side = weight-gradient GEMMs):
every combination of normal and high, in one process.  python tools/archive/ab_prio.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from fabric_amd import streams
torch.manual_seed(0)
dev = torch.device('cuda:0')
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
eng = model.engine()
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '?')
mk = lambda p: torch.cuda.Stream(device=dev, priority=p)
combos = {'chain high, side normal (shipped)': (mk(-1), mk(0)), 'chain normal, side normal': (mk(0), mk(0)),
          'chain high, side high': (mk(-1), mk(-1)), 'chain normal, side high': (mk(0), mk(-1))}
res = {k: [] for k in combos}
for rep in range(4):
    for name, (c, s) in combos.items():
        step._hp = c
        streams._streams[(dev.index or 0, 'wgrad')] = s
        with torch.cuda.stream(c):
            for _ in range(4): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(15): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 15)
for k, v in res.items(): print(f'{k:36s} median {statistics.median(v):.3f} ms/step', [round(x, 3) for x in v])
