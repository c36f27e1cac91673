"""Where does the host-fed step lose time?  (test infrastructure)  python tools/probe_feed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

def timed(fn, n=30, warm=6):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print('resident            ', round(timed(lambda i: ts.step(*dev[i % 3])), 3))
for depth in (2, 3):
    f = DeviceFeeder('cuda', depth=depth)
    def run(n):
        for b in f(pool[i % 3] for i in range(n)): ts.step(*b)
    run(6); torch.cuda.synchronize(); t = time.perf_counter(); run(30); torch.cuda.synchronize()
    print(f'feeder depth {depth}      ', round((time.perf_counter() - t) / 30 * 1e3, 3))
# background copies with no dependency at all: how much does a concurrent 110 MB H2D slow the step?
cs = torch.cuda.Stream()
tmp = [torch.empty_like(t, device='cuda') for t in pool[0]]
def bg(i):
    with torch.cuda.stream(cs):
        for d, h in zip(tmp, pool[i % 3]): d.copy_(h, non_blocking=True)
    ts.step(*dev[i % 3])
print('independent bg copy ', round(timed(bg), 3))
half = [tuple(t.to(torch.bfloat16).pin_memory() if t.dtype == torch.float32 else t for t in p) for p in pool]
tmph = [torch.empty_like(t, device='cuda') for t in half[0]]
def bgh(i):
    with torch.cuda.stream(cs):
        for d, h in zip(tmph, half[i % 3]): d.copy_(h, non_blocking=True)
    ts.step(*dev[i % 3])
print('independent bg copy, bf16 payload', round(timed(bgh), 3))
