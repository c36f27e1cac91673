"""In-process A/B of the wgrad grid-size knob (each setting gets a fresh TrainStep so workspaces are re-sized)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep
vals = [int(v) for v in sys.argv[1].split(',')]
key = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1 = BDN_TUNE_WGRAD_BLOCKS, 2 = BDN_TUNE_WGRAD_V3
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
res = {v: [] for v in vals}
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for v in vals:
        _lib.call('bdn_set_tuning', key, v)
        torch.manual_seed(0)
        model = BiDateNet(13, 2, precision='bf16').cuda().train()
        step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
        for _ in range(6): step.step(x1, x2, lbl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): step.step(x1, x2, lbl)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20)
        del step, model
for v in vals: print(f'tuning[{key}]={v}: median {statistics.median(res[v]):.3f} ms/step {res[v]}')
