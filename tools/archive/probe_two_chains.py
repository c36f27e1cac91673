"""Would two independent half-batch chains overlap better than one full-batch chain?  (test infrastructure)
arg: one64 | one32 | two32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
def mk(B, seed):
    torch.manual_seed(seed)
    m = BiDateNet(13, 2).cuda().train(); ts = TrainStep(m, lr=1e-3)
    x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
    lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
    ts.stream(); m.engine()._side_stream(torch.device('cuda', 0))      # take the streams now, in creation order
    return ts, (x1, x2, lbl)
which = sys.argv[1]
steps = {'one64': lambda: [mk(64, 0)], 'one32': lambda: [mk(32, 0)], 'two32': lambda: [mk(32, 1), mk(32, 2)]}[which]()
def run(n=30):
    for _ in range(5):
        for ts, d in steps: ts.step(*d)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        for ts, d in steps: ts.step(*d)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tot = sum(d[0].shape[0] for _, d in steps)
for rep in range(2):
    t = run(); print(which, round(t, 3), 'ms ->', round(tot / t * 1e3), 'pairs/s')
