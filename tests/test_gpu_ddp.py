"""-m gpu: the data-parallel step end to end on real kernels.  A gpurun box has ONE GPU, and RCCL refuses two
ranks on one device, so the two ranks share cuda:0 and exchange gradients over gloo; everything else (sharded
batch, local BatchNorm statistics, bucketed all-reduce launched from backward next to the wgrad stream, averaged
SGD update) is the code path the 8-GPU RCCL run takes.  With two or more GPUs visible the same worker also runs
over RCCL (backend 'nccl'), one rank per device.

Cases: the fp32 parity setting on 3 bands, and the DEFAULT bf16 setting on 13 bands -- the only configuration in
which the first conv's weight gradient takes the fused path whose final bucket is released from the main stream
(reference utils/helpers.py:333-335 is what this replaces).  The 'delay' variants park the weight-gradient stream
behind a long sleep kernel before every step: a bucket launched without being ordered behind that stream would
all-reduce stale gradients and the ranks would diverge."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
rank, world = int(sys.argv[1]), int(sys.argv[2])
precision, c, delay, backend = sys.argv[5], int(sys.argv[6]), sys.argv[7] == '1', sys.argv[8]
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[4]
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
devid = rank if backend == 'nccl' else 0
torch.cuda.set_device(devid)
dist.init_process_group(backend, rank=rank, world_size=world,
                        **({'device_id': torch.device('cuda', devid)} if backend == 'nccl' else {}))
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep
from oracle import filler
b, s, lr = 4, 32, 0.05
x1, x2, lbl = filler.make_inputs(b * world, c, s, seed=11)
x1, x2, lbl = (torch.from_numpy(v).cuda() for v in (x1, x2, lbl))
sl = slice(rank * b, (rank + 1) * b)                      # this rank's shard of the global batch
model = filler.fill_module(BiDateNet(c, 2, precision=precision)).cuda().train()
ts = TrainStep(model, lr=lr, n_buckets=3)
assert ts.world == world
eng = model.engine()
if precision == 'bf16' and c == 13:
    assert _lib.load().bdn_conv3x3_wgrad_bnbwd_supported(eng.dt, 2 * b, s, s, 64, 16, b), \
        'this case must exercise the fused first-layer weight gradient'
    last = ts.bucketer.buckets[-1][2]
    assert any(k.startswith('inc.conv.conv.3') for k in last) or any(k.startswith('down1') for k in last), last
side = eng._side_stream(torch.device('cuda', devid))
for _ in range(2):
    if delay:                                             # the weight-gradient stream is busy for ~20 ms (at least ten steps) when backward starts
        with torch.cuda.stream(side):
            torch.cuda._sleep(40_000_000)
    ts.step(x1[sl], x2[sl], lbl[sl])
torch.cuda.synchronize()
flat = ts.flat_params.cpu()
# every rank must hold identical parameters after the averaged update
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat.cuda() if backend == 'nccl' else flat)
assert all(torch.equal(o.cpu(), flat) for o in others), 'ranks diverged'
# single-process emulation of the same two DDP steps: per-shard gradients from identical weights, summed in rank
# order and applied by the same SGD kernel (so the comparison is exact up to the all-reduce's summation order)
models = [filler.fill_module(BiDateNet(c, 2, precision=precision)).cuda().train() for _ in range(world)]
steps = [TrainStep(m, lr=0.0, distributed=False) for m in models]   # lr 0, no communication: local gradients only
cur = steps[0].flat_params.clone()
for _ in range(2):
    g = torch.zeros_like(cur)
    for r, st in enumerate(steps):
        st.flat_params.copy_(cur)
        st.model.engine().invalidate_weights()
        st.step(x1[r * b:(r + 1) * b], x2[r * b:(r + 1) * b], lbl[r * b:(r + 1) * b])
        g += st.flat_grads
    torch.cuda.synchronize()
    _lib.call('bdn_sgd_step', cur.data_ptr(), g.data_ptr(), lr, 1.0 / world, cur.numel(), _lib.stream_ptr())
torch.cuda.synchronize()
err = (cur.cpu() - flat).abs().max().item()
assert err < 5e-6, err
dist.barrier(); dist.destroy_process_group()
print('ok', rank, err)
'''


def _run(tmp_path, precision, channels, delay, backend='gloo'):
    script = tmp_path / 'ddp_worker.py'
    script.write_text(_WORKER)
    port = str(31000 + (os.getpid() * 7 + channels + 2 * int(delay) + {'fp32': 0, 'bf16': 5, 'bf16x3': 11}[precision]) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', ROOT, port, precision, str(channels),
                               '1' if delay else '0', backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    assert all('ok' in o for o in outs)


def test_two_ranks_on_one_gpu_match_the_averaged_gradient_update(tmp_path):
    _run(tmp_path, 'fp32', 3, False)


@pytest.mark.parametrize('delay', [False, True])
def test_bf16_13band_two_ranks_fused_first_wgrad_path(tmp_path, delay):
    """The configuration the 8-GPU run uses (bf16, 13 bands): rank-identical parameters equal to the shard average,
    also when the weight-gradient stream lags far behind the main stream."""
    _run(tmp_path, 'bf16', 13, delay)


def test_fp32_two_ranks_delayed_side_stream(tmp_path):
    _run(tmp_path, 'fp32', 3, True)


def test_bf16x3_two_ranks(tmp_path):
    """The split-operand setting: weight gradients on the second stream with their own operand-split buffers, like bf16."""
    _run(tmp_path, 'bf16x3', 13, False)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank')
@pytest.mark.parametrize('delay', [False, True])
def test_bf16_13band_two_ranks_over_rccl(tmp_path, delay):
    _run(tmp_path, 'bf16', 13, delay, backend='nccl')


_RCCL1 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
from fabric_amd.parallel import init_rccl
init_rccl(0, 1, torch.device('cuda', 0))          # the helper bench.py / train.py use
from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from oracle import filler
delay = sys.argv[3] == '1'
b, c, s = 8, 13, 64
x1, x2, lbl = (torch.from_numpy(v).cuda() for v in filler.make_inputs(b, c, s, seed=21))
res = []
for force in (False, True):
    model = filler.fill_module(BiDateNet(c, 2, precision='bf16')).cuda().train()
    ts = TrainStep(model, lr=0.05, force_collectives=force, guard=False)      # the all-reduce calls are COUNTED below: no guard steps in between
    assert ts.bucketer.active() == force and len(ts.bucketer.buckets) == 5
    launched = []
    if force:
        orig = dist.all_reduce
        def counting(t, *a, **k):
            launched.append(t.numel())
            return orig(t, *a, **k)
        dist.all_reduce = counting
    side = model.engine()._side_stream(torch.device('cuda', 0))
    with torch.cuda.stream(ts.stream()):
        for _ in range(4):
            if delay:                       # weight-gradient stream ~20 ms behind: a bucket not ordered behind it would reduce stale gradients
                with torch.cuda.stream(side):
                    torch.cuda._sleep(40_000_000)
            ts.step(x1, x2, lbl)
    torch.cuda.synchronize()
    if force:
        dist.all_reduce = orig
        assert len(launched) == 4 * 5 and sum(launched[:5]) == ts.bucketer.reduce_end, launched[:5]
    res.append(ts.flat_params.cpu().clone())
assert torch.equal(res[0], res[1]), (res[0] - res[1]).abs().max()      # a sum over one rank changes nothing, in any order
dist.barrier(); dist.destroy_process_group()
print('ok')
'''


@pytest.mark.parametrize('delay', [False, True])
def test_bucket_allreduces_through_rccl_with_one_rank(tmp_path, delay):
    """RCCL on the real stream arrangement with the one GPU a test box has: process group 'nccl', world size 1, the five
    bucket all-reduces FORCED (TrainStep(force_collectives=True)) so that every step initialises / launches / orders them
    exactly as an 8-GPU run does -- from the weight-gradient stream, and the last bucket from the chain's stream after it
    joined the weight-gradient stream.  Four steps must not deadlock and must leave the parameters bit-identical to the
    purely local step, also with the weight-gradient stream parked behind a 20 ms sleep kernel before every step."""
    script = tmp_path / 'rccl1_worker.py'
    script.write_text(_RCCL1)
    port = str(33000 + (os.getpid() * 3 + int(delay)) % 2000)
    p = subprocess.Popen([sys.executable, str(script), ROOT, port, '1' if delay else '0'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.communicate(timeout=280)[0].decode()
    assert p.returncode == 0 and 'ok' in out, out


_GUARD = r'''
import json, os, sys, torch
sys.path.insert(0, sys.argv[1])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
from fabric_amd import BiDateNet, streams
from fabric_amd.parallel import init_rccl
from fabric_amd.train_step import TrainStep
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').to(dev).train()
B = 64
g = torch.Generator(device='cpu').manual_seed(5)
x1 = torch.randn(B, 13, 128, 128, generator=g).to(dev); x2 = torch.randn(B, 13, 128, 128, generator=g).to(dev)
lbl = (torch.rand(B, 128, 128, generator=g) < 0.1).to(torch.uint8).to(dev)
local = TrainStep(model, lr=1e-3)
for _ in range(3):
    local.step(x1, x2, lbl)                                   # the chain / weight-gradient streams exist ...
torch.cuda.synchronize()
init_rccl(0, 1, dev, high_priority=True)                      # ... BEFORE a process group with a HIGH-priority collective stream
before = {k: v.clone() for k, v in model.state_dict().items()}
forced = TrainStep(model, lr=1e-3, force_collectives=True)    # guard=True: runs inside the first step()
chain0, wgrad0 = streams.get('chain', dev).cuda_stream, streams.get('wgrad', dev).cuda_stream
rep = forced.guard_collectives(B, 128, 128)
after = model.state_dict()
rep['state_restored'] = all(torch.equal(before[k], after[k]) for k in before)
rep['streams_replaced'] = [streams.get('chain', dev).cuda_stream != chain0, streams.get('wgrad', dev).cuda_stream != wgrad0]
forced.step(x1, x2, lbl); torch.cuda.synchronize()           # and the guarded arrangement still trains
rep['loss_finite'] = bool(torch.isfinite(forced.last_logits).all())
print('GUARD ' + json.dumps(rep))
import torch.distributed as dist
dist.destroy_process_group()
'''


def test_guard_recovers_the_slow_collective_stream_arrangement(tmp_path):
    """The stream arrangement that measured +48...+59 % step time in round 3 -- the step's streams created first, then a process
    group whose collective stream is HIGH priority -- is provoked on purpose (one rank, buckets forced through RCCL), and
    TrainStep.guard_collectives must (i) see it if it is there, (ii) end at most 5 % above the local step after its remedies (a new
    chain stream is what fixes it on the boxes measured), (iii) leave parameters and BatchNorm buffers exactly as they were."""
    import json
    script = tmp_path / 'guard_worker.py'
    script.write_text(_GUARD)
    port = str(35000 + os.getpid() % 2000)
    p = subprocess.Popen([sys.executable, str(script), ROOT, port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.communicate(timeout=280)[0].decode()
    assert p.returncode == 0, out
    rep = json.loads([l for l in out.splitlines() if l.startswith('GUARD ')][-1][6:])
    assert rep['state_restored'] and rep['loss_finite'], rep
    assert rep['ok'] and rep['overhead_frac'] <= 0.05, rep
    first = rep['tried'][0]['overhead_frac']
    print(f"provoked arrangement {first * 100:+.1f} % -> kept {rep['kept']!r} at {rep['overhead_frac'] * 100:+.1f} % after {[t['arrangement'] for t in rep['tried']]}")
    if first <= 0.05:
        # the provocation did not produce the slow state on THIS box: say so instead of passing silently -- the repair path was
        # not exercised by hardware here (its decision logic is held by test_guard_decisions_on_scripted_timings either way)
        pytest.skip(f'slow collective-stream arrangement NOT observed on this box (provoked overhead {first * 100:+.1f} %): repair path not exercised')
    assert rep['placement_problem'] and rep['recovered'], rep
    assert any(rep['streams_replaced']) or rep['deferred_buckets'], rep


_GUARD_SCRIPTED = r'''
import json, os, sys, warnings, torch
sys.path.insert(0, sys.argv[1])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
from fabric_amd import BiDateNet, streams
from fabric_amd.parallel import init_rccl
from fabric_amd.train_step import TrainStep
init_rccl(0, 1, dev)
torch.manual_seed(0)
model = BiDateNet(3, 2, precision='bf16').to(dev).train()
out = {}

def scenario(name, table, boom=False, exchange=0.05e-3):
    """table(enabled, defer, chain_replaced, wgrad_replaced) -> seconds per step, exchange = seconds of the bucket all-reduces alone:
    the guard's clocks, scripted"""
    ts = TrainStep(model, lr=1e-3, force_collectives=True, guard=False)
    chain0, wgrad0 = streams.get('chain', dev), streams.get('wgrad', dev)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    calls = []
    def fake(x1, x2, lbl, n, warm):
        key = (ts.bucketer.enabled, ts.bucketer.defer, streams.get('chain', dev).cuda_stream != chain0.cuda_stream,
               streams.get('wgrad', dev).cuda_stream != wgrad0.cuda_stream)
        calls.append(key)
        if boom and len(calls) == 2:
            raise RuntimeError('scripted failure')
        ts._step(x1, x2, lbl)                      # one real step, so that the restore has something to undo
        return table(*key)
    ts._time_steps = fake
    ts._time_exchange = lambda n: exchange
    rec = {}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        try:
            rec['rep'] = ts.guard_collectives(4, 32, 32)
        except RuntimeError as e:
            rec['raised'] = str(e)
        rec['warned'] = [str(x.message)[:60] for x in w if issubclass(x.category, RuntimeWarning)]
    torch.cuda.synchronize()
    after = model.state_dict()
    rec['state_restored'] = all(torch.equal(before[k], after[k]) for k in before)
    rec['chain_is_original'] = streams.get('chain', dev).cuda_stream == chain0.cuda_stream
    rec['wgrad_is_original'] = streams.get('wgrad', dev).cuda_stream == wgrad0.cuda_stream
    rec['defer'] = bool(ts.bucketer.defer)
    rec['enabled'] = bool(ts.bucketer.enabled)
    rec['report_attr'] = ts.collectives_report if not isinstance(ts.collectives_report, dict) else 'dict'
    rec['calls'] = len(calls)
    ts.step(torch.randn(4, 3, 32, 32, device=dev), torch.randn(4, 3, 32, 32, device=dev), torch.zeros(4, 32, 32, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    streams.restore('chain', chain0, dev); streams.restore('wgrad', wgrad0, dev)       # next scenario starts from the same arrangement
    out[name] = rec

ms = 1e-3
# A: the overhead is explained by the exchange alone (0.8 ms of all-reduces): nothing may change, nothing may be warned about
scenario('exchange', lambda en, de, c, w: 6.0 * ms if not en else (7.0 * ms if de else 6.6 * ms), exchange=0.8 * ms)
# B: placement problem that a new chain stream fixes
scenario('fixed_by_chain', lambda en, de, c, w: 6.0 * ms if not en else (6.3 * ms if de else (6.05 * ms if c else (8.9 * ms if w else 9.0 * ms))))
# C: placement problem that the stream remedies make worse: back to the ORIGINAL streams, buckets deferred, overhead = deferred's
scenario('only_defer_helps', lambda en, de, c, w: 6.0 * ms if not en else (6.6 * ms if de else (9.6 * ms if c else (9.5 * ms if w else 9.0 * ms))))
# D: the measurement itself fails
scenario('failure', lambda en, de, c, w: 6.0 * ms, boom=True)
print('SCRIPTED ' + json.dumps(out))
import torch.distributed as dist
dist.destroy_process_group()
'''


def test_guard_decisions_on_scripted_timings(tmp_path):
    """TrainStep.guard_collectives' decision logic with its clock scripted (the hardware state it reacts to cannot be ordered up):
    an overhead that is the exchange itself changes nothing and warns about nothing; a placement problem is repaired by the stream
    remedy that measures best; remedies that measure worse are rolled back (streams.restore) before the buckets are deferred, and the
    reported overhead is that of the arrangement actually kept; a failing measurement leaves no half-state behind."""
    import json
    script = tmp_path / 'guard_scripted.py'
    script.write_text(_GUARD_SCRIPTED)
    port = str(37000 + os.getpid() % 2000)
    p = subprocess.Popen([sys.executable, str(script), ROOT, port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    txt = p.communicate(timeout=280)[0].decode()
    assert p.returncode == 0, txt
    out = json.loads([l for l in txt.splitlines() if l.startswith('SCRIPTED ')][-1][9:])
    a, b, c, d = out['exchange'], out['fixed_by_chain'], out['only_defer_helps'], out['failure']
    for r in (a, b, c, d):
        assert r['state_restored'] and r['enabled'], r
    ra = a['rep']
    assert ra['placement_problem'] is False and ra['kept'] == 'original' and ra['ok'] and not ra['deferred_buckets'], ra
    assert abs(ra['overhead_frac'] - 0.10) < 1e-6 and a['chain_is_original'] and a['wgrad_is_original'] and not a['warned'] and not a['defer'], a
    rb = b['rep']
    assert rb['placement_problem'] and rb['recovered'] and rb['ok'] and rb['kept'] == 'new_chain_stream' and not rb['deferred_buckets'], rb
    assert abs(rb['overhead_frac'] - 0.05 / 6.0) < 1e-6 and not b['chain_is_original'] and not b['warned'], b
    rc = c['rep']
    assert rc['placement_problem'] and rc['kept'] == 'original + deferred_buckets' and rc['deferred_buckets'] and c['defer'], rc
    assert c['chain_is_original'] and c['wgrad_is_original'], c                      # the remedies that measured worse were rolled back
    assert abs(rc['overhead_frac'] - 0.10) < 1e-6 and not rc['ok'] and len(c['warned']) == 1, c      # reports what runs; still above 5 %: says so
    assert d.get('raised') == 'scripted failure' and d['report_attr'] is None and not d['defer'], d


def test_bench_two_ranks_end_to_end_on_one_device():
    """bench.py's N > 1 code path for real: torch.distributed.run with two ranks (the driver's command line), both on cuda:0 with gloo
    in place of RCCL (BENCH_BACKEND=gloo: the one thing a single-GPU box cannot do is give each rank its own device).  One JSON line
    from rank 0 with n_gpus == 2, weak scaling, whole-job value = 2 ranks x batch x steps / window, three windows."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BENCH_BACKEND'] = 'gloo'
    port = str(36000 + os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port', port,
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '8', '--size', '64']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 3 and len(d['windows_ms_per_step']) == 3
    assert d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 2 * 8 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value'] and d['value'] > 0
    assert d['roofline'] is not None and d['step_classes'] is not None and 'cpu_baseline' not in d


def test_bench_eight_ranks_rehearsal_on_one_device():
    """BASELINE configs[2]'s launch -- `bench.py --gpus 8` under torch.distributed.run -- rehearsed with the one device a box has: eight
    ranks share cuda:0 and exchange gradients over gloo (BENCH_BACKEND=gloo; RCCL refuses several ranks per device).  It exercises
    everything of the 8-GPU run that is not the xGMI transfer itself: rank plumbing, the barrier / max-over-ranks timing, eight
    bucketers cutting and launching the same five buckets, the broadcast, the JSON contract with n_gpus = 8 and global batch 8 x B.
    Still "unmeasured on hardware" for scaling: this removes launch-logic surprises from the first real 8-GPU run, nothing more."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BENCH_BACKEND'] = 'gloo'
    env['OMP_NUM_THREADS'] = '2'
    port = str(38000 + os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1', '--master-port', port,
           os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '4', '--size', '64', '--windows', '2']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['steps'] == 2 and len(d['windows_ms_per_step']) == 2
    assert d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp8' and 'configs[2]' in d['config']['workload']
    assert abs(d['value'] - 8 * 4 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value'] and d['value'] > 0
    assert d['roofline'] is not None and 'families' in d['roofline'] and 'cpu_baseline' not in d
    assert np.isfinite(d['final_loss'])
