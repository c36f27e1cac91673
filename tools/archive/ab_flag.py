"""In-process A/B of a boolean / integer engine attribute:  python tools/archive/ab_flag.py pair_wgrad_handoff 0 1"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
typ = type(getattr(model.engine(), name))
for kv in os.environ.get('AB_SET', '').split(','):          # other engine attributes held fixed during the A/B:  AB_SET=wgrad_blocks=192
    if kv:
        k, v = kv.split('=')
        setattr(model.engine(), k, type(getattr(model.engine(), k))(int(v)))
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        setattr(model.engine(), name, typ(v))
        for _ in range(5): step.step(x1, x2, lbl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): step.step(x1, x2, lbl)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20)
for v in vals: print(f'{name}={v}: median {statistics.median(res[v]):.3f} ms/step {[round(t, 3) for t in res[v]]}')
