"""-m gpu: the data-parallel step end to end on real kernels.  A gpurun box has ONE GPU, and RCCL refuses two
ranks on one device, so the two ranks share cuda:0 and exchange gradients over gloo; everything else (sharded
batch, local BatchNorm statistics, bucketed all-reduce launched from backward next to the wgrad stream, averaged
SGD update) is the code path the 8-GPU RCCL run takes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[4]
dist.init_process_group('gloo', rank=rank, world_size=world)
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from oracle import filler
torch.cuda.set_device(0)
c, b, s, lr = 3, 4, 32, 0.05
x1, x2, lbl = filler.make_inputs(b * world, c, s, seed=11)
x1, x2, lbl = (torch.from_numpy(v).cuda() for v in (x1, x2, lbl))
sl = slice(rank * b, (rank + 1) * b)                      # this rank's shard of the global batch
model = filler.fill_module(BiDateNet(c, 2, precision='fp32')).cuda().train()
ts = TrainStep(model, lr=lr, n_buckets=3)
assert ts.world == world
for _ in range(2):
    ts.step(x1[sl], x2[sl], lbl[sl])
torch.cuda.synchronize()
flat = ts.flat_params.cpu()
# every rank must hold identical parameters after the averaged update
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
assert all(torch.equal(o, flat) for o in others), 'ranks diverged'
# single-process emulation of the same two DDP steps: per-shard gradients from identical weights, averaged
models = [filler.fill_module(BiDateNet(c, 2, precision='fp32')).cuda().train() for _ in range(world)]
steps = [TrainStep(m, lr=0.0, distributed=False) for m in models]   # lr 0, no communication: local gradients only
cur = steps[0].flat_params.clone()
for _ in range(2):
    g = torch.zeros_like(cur)
    for r, st in enumerate(steps):
        st.flat_params.copy_(cur)
        st.model.engine().invalidate_weights()
        st.step(x1[r * b:(r + 1) * b], x2[r * b:(r + 1) * b], lbl[r * b:(r + 1) * b])
        g += st.flat_grads
    cur = cur - lr * g / world
torch.cuda.synchronize()
err = (cur.cpu() - flat).abs().max().item()
assert err < 5e-6, err
dist.barrier(); dist.destroy_process_group()
print('ok', rank, err)
'''


def test_two_ranks_on_one_gpu_match_the_averaged_gradient_update(tmp_path):
    script = tmp_path / 'ddp_worker.py'
    script.write_text(_WORKER)
    port = str(31000 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', ROOT, port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    assert all('ok' in o for o in outs)
