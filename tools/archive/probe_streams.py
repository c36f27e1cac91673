"""Does the step time depend on WHICH pool streams carry the chain / the weight gradients?  (test infrastructure)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from fabric_amd import streams
B = 64
torch.manual_seed(0)
model = BiDateNet(13, 2).cuda().train()
ts = TrainStep(model, lr=1e-3)
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
eng = model.engine()
def t_res(n=30):
    with torch.cuda.stream(ts.stream()):
        for i in range(4): ts.step(x1, x2, lbl)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n): ts.step(x1, x2, lbl)
        torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 3)
print('default', t_res(), t_res())
key = str(x1.device)
res = []
for k in range(12):                      # the next normal-priority pool streams as the weight-gradient stream
    s = torch.cuda.Stream()
    streams._streams[(x1.device.index or 0, 'wgrad')] = s
    res.append((k, hex(s.cuda_stream)[-6:], t_res()))
print('side stream candidates', res)
best = min(res, key=lambda r: r[2])
res = []
for k in range(6):                       # the next high-priority pool streams as the chain stream
    ts._hp = torch.cuda.Stream(priority=-1)
    res.append((k, t_res()))
print('chain stream candidates', res)
