import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
def run(prio):
    m = BiDateNet(13, 2).cuda().train(); ts = TrainStep(m)
    B = 64
    x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn_like(x1); l = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
    st = torch.cuda.Stream(priority=-1) if prio else torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(8): ts.step(x1, x2, l)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): ts.step(x1, x2, l)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3
for p in (0, 1, 0, 1): print('high-priority main' if p else 'default main     ', f'{run(p):.3f} ms/step')
