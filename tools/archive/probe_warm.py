"""Does the step time drift over the life of a process?  (test infrastructure)"""
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
out = []
for blk in range(40):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(25): ts.step(x1, x2, lbl)
    torch.cuda.synchronize(); out.append(round((time.perf_counter() - t) / 25 * 1e3, 3))
print(out)
