"""In-process A/B of several engine configurations (test infrastructure):
    python tools/ab_cfg.py base: two:fwd_chains=2 two_l2:fwd_chains=2,fwd_chain_levels=2 fold:fold_bn_bwd=e1b+d4a+d3a
Each argument is  name:attr=value,attr=value  (engine attributes; ints are cast, fold_bn_bwd=e1b+d4a becomes a tuple, anything else stays a string).  The configurations
are interleaved, 4 rounds x 20 steps each after 5 warm-up steps; prints the median ms/step of each and the ratio to the first."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
cfgs = []
for a in sys.argv[1:]:
    name, _, rest = a.partition(':')
    kv = {}
    for item in filter(None, rest.split(',')):
        k, v = item.split('=')
        kv[k] = int(v) if v.lstrip('-').isdigit() else (tuple(x for x in v.split('+') if x) if (k == 'fold_bn_bwd' or '+' in v) else v)
    cfgs.append((name, kv))
B = int(os.environ.get('AB_BATCH', '64'))
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision=os.environ.get('AB_PRECISION', 'bf16')).cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
eng = model.engine()
defaults = {k: getattr(eng, k) for _, kv in cfgs for k in kv}
res = {n: [] for n, _ in cfgs}
with torch.cuda.stream(step.stream()):
    for rep in range(int(os.environ.get('AB_ROUNDS', '4'))):
        for name, kv in cfgs:
            for k, v in defaults.items(): setattr(eng, k, v)
            for k, v in kv.items(): setattr(eng, k, v)
            for _ in range(5): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20)
base = statistics.median(res[cfgs[0][0]])
for name, kv in cfgs:
    m = statistics.median(res[name])
    print(f'{name:28s} median {m:.3f} ms/step ({(m / base - 1) * 100:+.2f} %)  {[round(t, 3) for t in res[name]]}  {kv}')
