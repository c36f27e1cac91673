"""Does hipGraph capture of the whole training step change its duration?  (test infrastructure)  python tools/archive/probe_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
B = 64
torch.manual_seed(0)
model = BiDateNet(13, 2).cuda().train()
ts = TrainStep(model, lr=1e-3)
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.cuda.set_stream(ts.stream())

def timed(fn, n=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print('eager  ms/step', round(timed(lambda: ts.step(x1, x2, lbl)), 3))
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=ts.stream()):
        loss = ts._step(x1, x2, lbl)
    torch.cuda.synchronize()
    print('graph  ms/step', round(timed(lambda: g.replay()), 3), 'loss', float(loss))
    ref = float(ts.step(x1, x2, lbl))
    print('eager loss after replays', ref)
except Exception as e:
    print('capture failed:', repr(e)[:400])
