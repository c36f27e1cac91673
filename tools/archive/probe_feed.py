"""Host-fed vs resident step time, alternated inside one process (test infrastructure).  python tools/archive/probe_feed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from fabric_amd.input_pipeline import DeviceFeeder
B = 64
torch.manual_seed(0)
model = BiDateNet(13, 2).cuda().train()
ts = TrainStep(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
pool = [(torch.randn(B, 13, 128, 128, generator=g).pin_memory(), torch.randn(B, 13, 128, 128, generator=g).pin_memory(),
         (torch.rand(B, 128, 128, generator=g) < 0.1).to(torch.uint8).pin_memory()) for _ in range(3)]
dev = [tuple(t.cuda() for t in p) for p in pool]
torch.cuda.set_stream(ts.stream())
N = 60
def res():
    for i in range(5): ts.step(*dev[i % 3])
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(N): ts.step(*dev[i % 3])
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / N * 1e3, 3)
def fed(f):
    for b in f(pool[i % 3] for i in range(5)): ts.step(*b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for b in f(pool[i % 3] for i in range(N)): ts.step(*b)
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / N * 1e3, 3)
f3, f2 = DeviceFeeder('cuda', depth=3), DeviceFeeder('cuda', depth=2)
for rep in range(3):
    print('resident', res(), '| fed depth 3', fed(f3), '| fed depth 2', fed(f2), '| resident', res())
