"""Trace the slow and the fast state in one process (run under rocprofv3 --kernel-trace).  Markers: a 1-element fill kernel
(FillFunctor<int>) before each phase: 10 slow steps, trigger (5 extra streams + extra copy stream + depth-6 feeder), 10 fast steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
import fabric_amd.input_pipeline as ip
B = 64
torch.manual_seed(0)
model = BiDateNet(13, 2).cuda().train()
ts = TrainStep(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
pool = [(torch.randn(B, 13, 128, 128, generator=g).pin_memory(), torch.randn(B, 13, 128, 128, generator=g).pin_memory(),
         (torch.rand(B, 128, 128, generator=g) < 0.1).to(torch.uint8).pin_memory()) for _ in range(3)]
dev = [tuple(t.cuda() for t in p) for p in pool]
torch.cuda.set_stream(ts.stream())
mark = torch.zeros(1, dtype=torch.int32, device='cuda')
import subprocess, threading
def clocks(tag):
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    keep = [l.split(':', 1)[1].strip() if 'level' in l or 'Power' in l else '' for l in out.splitlines() if 'GPU[0]' in l]
    print(tag, ' | '.join(k for k in keep if k), flush=True)
def t_res(n=10, tag=None):
    for i in range(5): ts.step(*dev[i % 3])
    torch.cuda.synchronize(); mark.fill_(1); t = time.perf_counter()
    if tag:
        n = 400
        th = threading.Timer(1.0, clocks, args=(tag,)); th.start()
    for i in range(n): ts.step(*dev[i % 3])
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 3)
a = t_res(tag='slow-state clocks:')
streams = [torch.cuda.Stream() for _ in range(5)]
for s in streams:
    with torch.cuda.stream(s): torch.zeros(16, device='cuda').add_(1)
cs = torch.cuda.Stream()
f = ip.DeviceFeeder('cuda', depth=6)
for bb in f(pool[i % 3] for i in range(30)): ts.step(*bb)
torch.cuda.synchronize()
print('slow', a, 'fast', t_res(tag='fast-state clocks:'), t_res())
