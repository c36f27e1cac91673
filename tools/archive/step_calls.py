"""Per-entry-point time of one instrumented training step (test infrastructure):  python tools/archive/step_calls.py [precision]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = x1 + 0.3 * torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision=prec).cuda().train()
step = TrainStep(model, lr=1e-3)
with torch.cuda.stream(step.stream()):
    for _ in range(5): step.step(x1, x2, lbl)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    step.step(x1, x2, lbl)
    torch.cuda.synchronize()
    raw, _lib.PROFILE = _lib.PROFILE, None
agg = collections.defaultdict(lambda: [0, 0.0])
for name, phase, h, e0, e1 in raw:
    a = agg[(name, phase)]; a[0] += 1; a[1] += e0.elapsed_time(e1)
for (name, phase), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{name:28s} {phase:4s} {n:3d} calls {ms:8.3f} ms')
