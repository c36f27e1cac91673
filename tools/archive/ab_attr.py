"""In-process A/B of a module-level switch of fabric_amd.engine (test infrastructure):  python tools/archive/ab_attr.py NAME v0 v1 ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet, engine
from fabric_amd.train_step import TrainStep
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3)
res = {v: [] for v in vals}
with torch.cuda.stream(step.stream()):
    for rep in range(5):
        for v in vals:
            setattr(engine, name, v)
            for _ in range(5): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 30)
for v in vals: print(f'{name}={v}: median {statistics.median(res[v]):.4f} ms/step {[round(t, 3) for t in res[v]]}')
