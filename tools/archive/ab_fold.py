"""In-process A/B of engine.fold_bn_bwd (BatchNorm backward applied inside the data-gradient conv):  python tools/archive/ab_fold.py "" d4a d4a,d3a,d3b ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
sets = [tuple(x for x in a.split(',') if x) for a in (sys.argv[1:] or ['', 'd4a'])]
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3)
res = {s: [] for s in sets}
with torch.cuda.stream(step.stream()):
    for rep in range(4):
        for s in sets:
            model.engine().fold_bn_bwd = s
            for _ in range(5): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
            res[s].append(e0.elapsed_time(e1) / 30)
for s in sets: print(f'fold={",".join(s) or "-":28s} median {statistics.median(res[s]):.4f} ms/step {[round(t, 3) for t in res[s]]}')
