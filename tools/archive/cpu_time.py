"""How long does the host need to enqueue one train step? (test infrastructure)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
m = BiDateNet(13, 2).cuda().train(); ts = TrainStep(m)
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn_like(x1); l = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
for _ in range(5): ts.step(x1, x2, l)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): ts.step(x1, x2, l)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue per step {(t1 - t0) / 10 * 1e3:.2f} ms ; total per step {(t2 - t0) / 10 * 1e3:.2f} ms')
