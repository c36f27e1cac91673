"""Provoke the slow RCCL stream arrangement (round 3: +48...+59 % step time) and watch TrainStep.guard_collectives deal with it.

    python tools/archive/probe_rccl_guard.py [slow|fast]

slow: the step's streams exist BEFORE the process group, whose collective stream is HIGH priority (the one combination that measured
slow); fast: group first, default priority (what bench.py / train.py do).  One rank, buckets forced.  Prints the guard's report."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from fabric_amd import BiDateNet, streams
from fabric_amd.parallel import init_rccl
from fabric_amd.train_step import TrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else 'slow'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
os.environ.setdefault('MASTER_PORT', str(29000 + os.getpid() % 3000))
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').to(dev).train()
if mode == 'fast':
    init_rccl(0, 1, dev)
local = TrainStep(model, lr=1e-3)
B = 64
x1 = torch.randn(B, 13, 128, 128, device=dev); x2 = torch.randn(B, 13, 128, 128, device=dev)
lbl = (torch.rand(B, 128, 128, device=dev) < 0.1).to(torch.uint8)
for _ in range(3):
    local.step(x1, x2, lbl)              # creates the chain / weight-gradient streams
torch.cuda.synchronize()
if mode != 'fast':
    init_rccl(0, 1, dev, high_priority=True)
forced = TrainStep(model, lr=1e-3, force_collectives=True, guard=False)
rep = forced.guard_collectives(B, 128, 128, verbose=True)
print(json.dumps(rep, indent=1))
print('parked streams:', len(streams._graveyard))
