"""CPU tests of the host side: C-ABI surface, module / state-dict schema, dataset API, DDP bucketer."""
import ctypes
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/bidate_hip.h <-> libbidate_hip.so <-> fabric_amd/_lib.py agree (no compute call: no GPU here)."""
    from fabric_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'bidate_hip.h')).read()
    declared = set(re.findall(r'\b(bdn_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert _lib.load().bdn_version() >= 1
    assert isinstance(_lib.load().bdn_last_error(), bytes)


def test_argument_validation_needs_no_gpu():
    from fabric_amd import _lib
    lib = _lib.load()
    assert lib.bdn_conv3x3_num_mtiles(128, 128, 128, 64, 64) == 128 * 8 * 8      # 16x16 tiles for 64-wide outputs
    assert lib.bdn_conv3x3_num_mtiles(128, 128, 128, 128, 64) == 128 * 16 * 8    # 8x16 tiles
    assert lib.bdn_conv3x3_num_mtiles(128, 8, 8, 512, 64) == 64                   # two 8x8 images per tile
    assert lib.bdn_conv3x3_num_mtiles_ex(_lib.BDN_BF16X3, 128, 128, 128, 64, 64, 64) == 128 * 16 * 8     # fused split product: no 16x16 tiles
    assert lib.bdn_conv3x3_num_mtiles_ex(_lib.BDN_BF16X3, 128, 128, 128, 16, 64, 64) == 128 * 16 * 8     # 16-channel operand: the same, 16-channel chunks
    assert lib.bdn_conv3x3_num_mtiles_ex(_lib.BDN_BF16, 128, 128, 128, 64, 64, 64) == 128 * 8 * 8
    assert lib.bdn_wgrad_workspace_bytes(2, 16, 16, 64, 64, 1) > 0
    with pytest.raises(RuntimeError, match='null pointer'):
        _lib.call('bdn_conv3x3', 1, None, 64, None, 0, 0, None, 1, None, None, None, None, 1, 8, 8, 64, None)
    with pytest.raises(RuntimeError, match='multiple of 64'):
        _lib.call('bdn_conv3x3', 1, 1, 64, None, 0, 0, None, 1, 1, None, 1, None, 1, 8, 8, 65, None)
    with pytest.raises(RuntimeError, match='4 GB'):       # 32-bit byte offsets inside the kernel: 4096 x 128 x 128 x 64 bf16 = 8 GB
        _lib.call('bdn_conv3x3', 1, 1, 64, None, 0, 0, None, 1, 1, None, 1, None, 4096, 128, 128, 64, None)
    # round-2 entry points: plans are pure functions of their arguments, errors come before anything touches a device
    from fabric_amd._lib import BDN_BF16, BDN_BF16X3, BDN_F32, IN_BNRELU, IN_PLAIN, WG_ROLE, WG_SIMPLE, wg_flags
    var = lib.bdn_conv3x3_wgrad_variant
    assert var(BDN_BF16, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, 0) == WG_ROLE       # role-split kernel, plain operands by LDS-DMA
    assert var(BDN_BF16, 128, 64, 64, 128, 128, 0, 64, IN_BNRELU, 0) == WG_ROLE      # ... BatchNorm+ReLU on load by its producer waves
    assert var(BDN_BF16, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, wg_flags(kernel=WG_SIMPLE)) == WG_SIMPLE
    assert var(BDN_F32, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, 0) == WG_SIMPLE
    assert var(BDN_BF16, 128, 8, 8, 512, 512, 0, 64, IN_PLAIN, 0) == WG_SIMPLE
    wsb = lib.bdn_wgrad_workspace_bytes_ex
    half = wsb(BDN_BF16, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, 0)                      # default grid: 128 blocks, 4 tiles -> 32 splits
    assert half == 32 * 9 * 128 * 128 * 4
    assert wsb(BDN_BF16, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, wg_flags(0, 0, 256)) == 2 * half
    assert wsb(BDN_BF16X3, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, 0) > half             # doubled operands + the quadrant tile
    assert lib.bdn_wgrad_workspace_bytes(128, 64, 64, 128, 128, 64) >= wsb(BDN_BF16X3, 128, 64, 64, 128, 128, 0, 64, IN_PLAIN, 0)
    assert lib.bdn_conv3d_num_mtiles(8, 5, 128, 128) == 8 * 5 * 16 * 8
    assert lib.bdn_conv3x3_variant(BDN_BF16, 128, 64, 64, 128, 0, 128, 64) == b'conv3x3_kernel<bf16,128,8,16,1,128,1,4,false,bf16,false,false,false,0,false>'
    assert lib.bdn_conv3x3_variant(BDN_BF16X3, 128, 64, 64, 128, 0, 128, 64).endswith(b'float,false,false,false,3,false>')
    assert lib.bdn_conv3x3_x3src_variant(BDN_BF16X3, 128, 64, 64, 128, 128, 64) == b'conv3x3_kernel<bf16,128,8,16,1,128,1,4,false,float,false,false,false,3,true>'
    assert lib.bdn_conv3x3_x3src_variant(BDN_BF16X3, 128, 64, 64, 1024, 128, 64) == b''        # operands beyond the staging table
    for name, args, msg in (
            ('bdn_conv3d', (BDN_BF16, None, 64, 0, None, 1, None, None, None, None, 1, 1, 8, 8, 64, None), 'null pointer'),
            ('bdn_conv3d', (7, 1, 64, 0, None, 1, 1, None, 1, None, 1, 1, 8, 8, 64, None), 'bad dtype'),
            ('bdn_conv3d', (BDN_BF16, 1, 60, 0, None, 1, 1, None, 1, None, 1, 1, 8, 8, 64, None), 'multiple'),
            ('bdn_conv3d_wgrad', (BDN_BF16, 1, 64, 1, 64, 1, 1, 80, 1, 1, 8, 8, None), 'Cin_real'),
            ('bdn_split_pack', (1, 12, None, 0, 0, None, 1, 1, 1, 8, 8, None), 'bad shape'),
            ('bdn_split_pack', (1, 16, None, 0, 1, None, 1, 1, 1, 8, 8, None), 'needs in_bn'),
            ('bdn_bnrelu', (BDN_BF16, 1, 1, 1, 1, 2, 8, 8, 24, None), 'bad shape'),
            ('bdn_conv3x3_wgrad_ex', (BDN_BF16X3, 1, 64, 1, 64, 1, 64, 0, None, 1, 1, 1, 64, 2, 8, 8, 3, None), 'one split-packed'),
            ('bdn_conv3x3_wgrad_ex', (BDN_BF16, 1, 64, 1, 64, None, 0, 0, None, 1, 1, 1, 64, 2, 8, 8, 0, None), 'phases')):
        with pytest.raises(RuntimeError, match=msg):
            _lib.call(name, *args)
    assert not hasattr(lib, 'bdn_set_tuning')                  # no process-wide tuning state any more


def test_state_dict_schema_matches_reference():
    """128 entries, reference key names and shapes."""
    from fabric_amd import BiDateNet
    from fabric_amd.engine import param_order
    from oracle.bidate_oracle import build_torch_baseline
    m = BiDateNet(13, 2)
    sd = m.state_dict()
    ref = build_torch_baseline(13, 2).state_dict()
    assert len(sd) == 128 and list(sd) == list(ref)
    assert all(sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 13401154
    assert sd['inc.conv.conv.0.weight'].shape == (64, 13, 3, 3)
    assert sd['up1.conv.conv.0.weight'].shape == (256, 1024, 3, 3)
    order = param_order(13)
    assert set(order) == {k for k, _ in m.named_parameters()} and len(order) == 74


def _reference_like_pickle(model, monkeypatch):
    """Bytes of torch.save(nn.DataParallel(model)) as the REFERENCE would have written them (train.py:222): the BiDateNet state
    holds sub-modules and parameters only -- none of the attributes this build adds (_engine, precision, n_channels, n_classes)."""
    import io
    from fabric_amd.models.bidate_model import BiDateNet
    added = ('_engine', 'precision', 'n_channels', 'n_classes')
    monkeypatch.setattr(BiDateNet, '__getstate__', lambda self: {k: v for k, v in self.__dict__.items() if k not in added})
    buf = io.BytesIO()
    torch.save(torch.nn.DataParallel(model), buf)
    monkeypatch.undo()
    return buf.getvalue()


def test_reference_whole_module_pickle_loads_into_a_usable_model(monkeypatch):
    """SURVEY.md 8b / reference train.py:222: the reference persists torch.save(DataParallel(BiDateNet)).  Unpickling resolves the class
    paths through the root `models.*` shims; __setstate__ rebuilds what the reference never had; load_checkpoint unwraps the
    DataParallel and returns a model whose engine() can be built."""
    import io
    import models.bidate_model as shim                        # the reference's import path
    from fabric_amd.utils.helpers import load_checkpoint
    from oracle import filler
    src = filler.fill_module(shim.BiDateNet(3, 2))
    blob = _reference_like_pickle(src, monkeypatch)
    assert b'_engine' not in blob and b'precision' not in blob and b'n_classes' not in blob
    obj = torch.load(io.BytesIO(blob), weights_only=False)
    inner = obj.module
    assert type(inner).__name__ == 'BiDateNet' and inner._engine is None
    assert (inner.n_channels, inner.n_classes, inner.precision) == (3, 2, os.environ.get('BIDATE_PRECISION', 'bf16'))
    assert inner.engine().layers[0].cin_real == 3              # the engine builds (raises if the HIP library is missing)
    m = load_checkpoint(io.BytesIO(blob), precision='fp32')
    assert (m.n_channels, m.n_classes, m.precision) == (3, 2, 'fp32') and not isinstance(m, torch.nn.DataParallel)
    sd, ref = m.state_dict(), src.state_dict()
    assert list(sd) == list(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
    # a 13-band pickle derives 13
    blob13 = _reference_like_pickle(shim.BiDateNet(13, 2), monkeypatch)
    assert torch.load(io.BytesIO(blob13), weights_only=False).module.n_channels == 13


def test_module_prefixed_state_dict_loads():
    """A state dict saved from the reference's DataParallel wrapper carries `module.` on every key (helpers.py:335): a plain
    load_state_dict rejects it, load_checkpoint takes it (and the bare and nested forms) and keeps the BatchNorm buffers."""
    from fabric_amd import BiDateNet
    from fabric_amd.utils.helpers import load_checkpoint, strip_module_prefix
    from oracle import filler
    src = filler.fill_module(BiDateNet(13, 2))
    ref = src.state_dict()
    prefixed = {'module.' + k: v.clone() for k, v in ref.items()}
    with pytest.raises(RuntimeError, match='Missing key|Unexpected key'):
        BiDateNet(13, 2).load_state_dict(prefixed)
    assert list(strip_module_prefix(prefixed)) == list(ref)
    for form in (prefixed, dict(ref), {'state_dict': prefixed}, {'model': torch.nn.DataParallel(src)}, src):
        m = load_checkpoint(form)
        sd = m.state_dict()
        assert m.n_channels == 13 and list(sd) == list(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
        assert sd['inc.conv.conv.1.num_batches_tracked'].dtype == torch.int64
    bad = dict(prefixed)
    del bad['module.up3.conv.conv.4.running_var']
    with pytest.raises(RuntimeError, match='running_var'):
        load_checkpoint(bad)
    with pytest.raises(RuntimeError, match='not a BiDateNet checkpoint'):
        load_checkpoint({'foo': torch.zeros(1)})


@pytest.mark.skipif(not os.path.isdir('/root/reference/models'), reason='the reference tree is only present in the build container')
def test_genuine_reference_pickle_loads(tmp_path):
    """The real thing where the reference is at hand: a separate interpreter imports the REFERENCE's BiDateNet, wraps it as
    utils/helpers.py:333-335 does and saves it as train.py:222 does; this process (repo root on sys.path) loads the file."""
    import subprocess
    from fabric_amd.utils.helpers import load_checkpoint
    out = tmp_path / 'checkpoint_epoch_0.pt'
    code = ('import sys, torch; sys.path.insert(0, "/root/reference"); torch.manual_seed(3);'
            'from models.bidate_model import BiDateNet; import torch.nn as nn;'
            f'm = nn.DataParallel(BiDateNet(3, 2)); torch.save(m, r"{out}"); torch.save(m.state_dict(), r"{out}.sd")')
    subprocess.run([sys.executable, '-c', code], check=True, cwd=str(tmp_path))
    obj = torch.load(str(out), weights_only=False)
    assert type(obj.module).__module__ == 'fabric_amd.models.bidate_model' and obj.module.engine() is not None
    m = load_checkpoint(str(out))
    sd = torch.load(str(out) + '.sd')
    assert all(k.startswith('module.') for k in sd)
    got = m.state_dict()
    assert len(got) == 128 and all(torch.equal(got[k], sd['module.' + k]) for k in got)
    m2 = load_checkpoint(str(out) + '.sd')
    assert all(torch.equal(m2.state_dict()[k], got[k]) for k in got)


def test_modules_refuse_cpu_and_have_no_fallback():
    from fabric_amd import BiDateNet
    from fabric_amd.models import unet_parts
    m = BiDateNet(3, 2)
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match='fused HIP stages'):
        unet_parts.double_conv(3, 64)(torch.zeros(1, 3, 8, 8))
    src = open(os.path.join(ROOT, 'fabric_amd', 'engine.py')).read() + open(os.path.join(ROOT, 'fabric_amd', '_lib.py')).read()
    assert 'oracle' not in src, 'the product must never import the oracle'


def test_patch_pair_dataset_matches_reference_fixture(golden_dir):
    """G7: OneraPreloader / onera_siamese_loader on a fixed random.seed reproduce the reference's outputs."""
    from fabric_amd.utils.dataloaders import OneraPreloader
    g = np.load(os.path.join(golden_dir, 'g7_loader.npz'))
    r = np.random.default_rng(7)
    data = {'cityA': {'images': r.standard_normal((2, 3, 40, 36)).astype(np.float32),
                      'labels': (r.uniform(0, 1, (40, 36)) < 0.2).astype(np.uint8)},
            'cityB': {'images': r.standard_normal((2, 3, 30, 50)).astype(np.float32),
                      'labels': (r.uniform(0, 1, (30, 50)) < 0.2).astype(np.uint8)}}
    meta = [['cityA', 0, 0], ['cityA', 16, 8], ['cityB', 4, 30], ['cityB', 10, 0], ['cityA', 20, 20]]
    random.seed(1234)
    ds = OneraPreloader('unused/', meta, data, 12, aug=True)
    assert np.array_equal(np.array([[m[1], m[2], 0 if m[0] == 'cityA' else 1] for m in ds.imgs]), g['order'])
    for i in range(len(ds)):
        a, b, l = ds[i]
        assert a.dtype == np.float32 and l.dtype == np.uint8 and a.shape == (3, 12, 12) and l.shape == (12, 12)
        assert np.array_equal(a, g[f'img1_{i}']) and np.array_equal(b, g[f'img2_{i}']) and np.array_equal(l, g[f'lbl_{i}'])


def test_patch_origin_rule():
    from fabric_amd.utils.dataloaders import metadata_from_shapes, patch_origins
    assert patch_origins(300, 260, 90, 180) == [[0, 0], [180, 0]]     # metadata.json defaults; j=180 does not fit in 260
    tr, va = metadata_from_shapes({'a': (100, 100), 'b': (200, 95)}, ['b'], 90, 90)
    assert tr == [['a', 0, 0]] and va == [['b', 0, 0], ['b', 90, 0]]


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from fabric_amd.parallel import FlatLayout, GradBucketer, ShardSampler, allreduce_mean_grads, shard_indices
from fabric_amd.engine import param_order
from fabric_amd import BiDateNet
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[4]
dist.init_process_group('gloo', rank=rank, world_size=world)
m = BiDateNet(3, 2)
order = param_order(3)
lay = FlatLayout([(k, p.shape) for k, p in m.named_parameters()], order)
flat = torch.zeros(lay.total)
tail = [k for k in order if k.endswith('.bias') and k.split('.')[-2] in ('0', '3')]
b = GradBucketer(lay, flat, n_buckets=4, keys_no_reduce=tail)
assert len(b.buckets) == 5 and b.buckets[0][0] == 0 and b.buckets[-1][1] == b.reduce_end
assert all(b.buckets[i][1] == b.buckets[i + 1][0] for i in range(4))                 # contiguous, no gap or overlap
assert (b.buckets[-1][1] - b.buckets[-1][0]) * 4 <= (1 << 20) < (b.buckets[-2][1] - b.buckets[-2][0]) * 4   # small tail
assert sum(len(ks) for _, _, ks in b.buckets) == len(order) - len(tail)
b4 = GradBucketer(lay, flat, n_buckets=4, keys_no_reduce=tail, tail_bytes=0)
assert len(b4.buckets) == 4
for step in range(2):
    for k in order:                       # "backward": gradients appear in layout order
        g = lay.view(flat, k)
        g.fill_(0.0 if k in tail else float(rank + 1) * (1 + step))
        b.on_ready([k])
    b.finish()
    expect = sum(r + 1 for r in range(world)) * (1 + step)
    for k in order:
        v = lay.view(flat, k)
        assert torch.all(v == (0.0 if k in tail else expect)), (k, v.flatten()[0].item(), expect)
idx = [shard_indices(103, r, world) for r in range(world)]
assert all(len(i) == 103 // world for i in idx) and len(set(sum(idx, []))) == (103 // world) * world
# the autograd loop's gradient exchange: a few packed buckets, every .grad ends up as the mean over the ranks
torch.manual_seed(5)
ref = [torch.randn(p.shape) for p in m.parameters()]
for p, r in zip(m.parameters(), ref):
    p.grad = r * (rank + 1)
nb = allreduce_mean_grads(m.parameters(), world, bucket_bytes=16 << 20)
assert 1 <= nb <= 5, nb                                   # 53.6 MB of gradients: four buckets, not 74 calls
mean = sum(r + 1 for r in range(world)) / world
for p, r in zip(m.parameters(), ref):
    assert torch.allclose(p.grad, r * mean, rtol=1e-6, atol=1e-7)
# epoch-aware shards: this rank's indices are disjoint from the other rank's, together they cover the list, and every rank
# derives them WITHOUT touching the global RNG state (rank 1 burns some draws first)
import random
if rank == 1:
    random.random(); torch.rand(3)
mine = []
for epoch in range(2):
    smp = ShardSampler(103, rank, world, seed=7)
    smp.set_epoch(epoch)
    mine.append(list(smp))
gathered = [None] * world
dist.all_gather_object(gathered, mine)
for epoch in range(2):
    shards = [g[epoch] for g in gathered]
    flat_idx = sum(shards, [])
    assert all(len(sh) == 103 // world for sh in shards)
    assert len(set(flat_idx)) == len(flat_idx) == (103 // world) * world          # disjoint; at most world - 1 items dropped
assert gathered[0][0] != gathered[0][1]                                            # reshuffled per epoch
dist.barrier(); dist.destroy_process_group()
print('BUCKETS', [(a, e) for a, e, _ in b.buckets])
print('ok', rank)
'''


@pytest.mark.parametrize('world', [2, 8])
def test_grad_bucketer_gloo(tmp_path, world):
    """N>1 path on CPU: `world` gloo ranks (2, and 8 = BASELINE configs[2]'s rank count), bucketed async all-reduce over the flat
    gradient buffer, stride-by-rank shards, epoch-aware ShardSampler (disjoint, covering, reshuffled per epoch), and the SAME five
    contiguous buckets on every rank and at every world size (what a rank launches must match what its peers launch)."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    port = str(29500 + (os.getpid() + 17 * world) % 2000)
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), ROOT, port], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    assert all('ok' in o for o in outs)
    cuts = {[l for l in o.splitlines() if l.startswith('BUCKETS')][-1] for o in outs}
    assert len(cuts) == 1, cuts
    # and they are the cuts a process computes with no group at all (world-size independent)
    from fabric_amd import BiDateNet
    from fabric_amd.engine import param_order
    from fabric_amd.parallel import FlatLayout, GradBucketer
    m, order = BiDateNet(3, 2), param_order(3)
    lay = FlatLayout([(k, p_.shape) for k, p_ in m.named_parameters()], order)
    tail = [k for k in order if k.endswith('.bias') and k.split('.')[-2] in ('0', '3')]
    b = GradBucketer(lay, torch.zeros(lay.total), n_buckets=4, keys_no_reduce=tail)
    assert cuts.pop() == 'BUCKETS ' + str([(a, e) for a, e, _ in b.buckets])


def test_inference_tiler_matches_golden(golden_dir):
    """fabric_amd.utils.inference._get_patches/_get_bands vs fixture G7 (captured from the reference's
    utils/inference.py:134-236) and vs the oracle on ragged / exact-multiple scene sizes."""
    from fabric_amd.utils import inference as inf
    from oracle import bidate_oracle as O
    g = np.load(os.path.join(golden_dir, 'g7_tiling.npz'))
    hs, ws, lc, lr, h, w = [int(v) for v in g['meta']]
    r = np.random.default_rng(7)          # same generator state as oracle/make_golden.py before the tiling case
    r.standard_normal((2, 3, 40, 36)); r.uniform(0, 1, (40, 36)); r.standard_normal((2, 3, 30, 50)); r.uniform(0, 1, (30, 50))
    arr = r.standard_normal((300, 260, 13)).astype(np.float32)
    tiles, hs2, ws2, lc2, lr2, h2, w2 = inf._get_patches(arr, patch_dim=128)
    assert (hs2, ws2, lc2, lr2, h2, w2) == (hs, ws, lc, lr, h, w)
    assert np.allclose(tiles.reshape(9, -1).astype(np.float64).sum(1), g['tile_checksums'])
    img = inf._get_bands(g['pred'].astype(np.float64), hs, ws, lc, lr, h, w, patch_size=128)
    assert img.dtype == np.float64 and np.array_equal(img.astype(np.uint8), g['stitched'])
    for (hh, ww, p) in [(70, 33, 16), (64, 96, 32), (40, 40, 40), (129, 257, 128)]:
        a = r.standard_normal((hh, ww, 4)).astype(np.float32)
        t, *meta = inf._get_patches(a, p)
        to, *metao = O.tile_scene(a, p)
        assert np.array_equal(t, to) and tuple(meta) == tuple(metao)
        pred = r.integers(0, 2, (t.shape[0], p, p)).astype(np.float64)
        assert np.array_equal(inf._get_bands(pred, *meta, patch_size=p), O.stitch_scene(pred, *metao, p))
    with pytest.raises(ValueError):
        inf.tile_origins(100, 300, 128)


def test_streams_and_feeder_need_a_device():
    """The product has no CPU path: asking for the step's streams (or a feeder) without a ROCm device fails loudly, and the
    stream / event entry points validate their arguments before touching HIP."""
    import torch
    from fabric_amd import _lib, streams
    with pytest.raises(ValueError):
        streams.get('compute')
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU path'):
            streams.get('chain')
        from fabric_amd.input_pipeline import DeviceFeeder
        with pytest.raises(RuntimeError):
            DeviceFeeder('cpu')
    for name, args, msg in (('bdn_stream_create', (0, None), 'null pointer'), ('bdn_stream_create', (7, None), 'null pointer'),
                            ('bdn_stream_destroy', (None,), 'null pointer'), ('bdn_event_create', (None,), 'null pointer'),
                            ('bdn_event_record', (None, None), 'null pointer'), ('bdn_stream_wait_event', (None, None), 'null pointer')):
        with pytest.raises(RuntimeError, match=msg):
            _lib.call(name, *args)


def test_bench_launches_its_own_ranks_or_refuses():
    """`python bench.py --gpus N` without a launcher above it must become N ranks (torch.distributed.run on 127.0.0.1) or refuse with
    a non-zero exit -- never print a line whose n_gpus differs from --gpus (round-3 review: it silently ran ONE rank).  The launcher
    logic runs for real here with the compute replaced by a gloo head count on the CPU (--launcher-selftest)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    argv = ['--gpus', '2', '--steps', '3']
    assert bench.launch_plan(1, {}, 0, argv) == ('run', None)
    assert bench.launch_plan(2, {'WORLD_SIZE': '2'}, 0, argv) == ('run', None)            # we ARE a rank of a 2-rank launch
    assert bench.launch_plan(8, {'WORLD_SIZE': '1'}, 8, argv)[0] == 'refuse'               # a launcher that disagrees with --gpus
    assert bench.launch_plan(8, {}, 1, argv)[0] == 'refuse'                                # one visible device, eight asked for
    what, cmd = bench.launch_plan(2, {}, 2, argv)
    assert what == 'spawn' and '--nproc-per-node=2' in cmd and '127.0.0.1' in cmd and cmd[-len(argv):] == argv
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BENCH_ASSUME_DEVICES'] = '2'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher-selftest'],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['world_size_env'] == 2, r.stdout
    # no launcher and too few devices (this container has none): refusal, exit code 2, nothing on stdout
    env.pop('BENCH_ASSUME_DEVICES')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 2 and 'refusing' in r.stderr and not r.stdout.strip()


class _DrawDataset(torch.utils.data.Dataset):
    """Returns the augmentation draw itself: what utils/dataloaders.py:150-156 consumes from the global `random`."""

    def __len__(self):
        return 16

    def __getitem__(self, i):
        return torch.tensor([random.random(), random.random()], dtype=torch.float64)


def test_augmentation_draws_differ_between_ranks_with_loader_workers():
    """torch.manual_seed(seed) is the same on every rank, and loader workers seed `random` from the loader's base seed: without a
    per-rank generator / worker_init_fn every rank would draw the same flips and rotations.  make_loaders' loader
    arguments, on a dataset that returns the draws."""
    from fabric_amd.train import _RankWorkerSeed

    def draws(rank, epochs=2):
        torch.manual_seed(7)                                   # what train.py does on every rank
        gen = torch.Generator()
        gen.manual_seed(7 * 7919 + rank)
        dl = torch.utils.data.DataLoader(_DrawDataset(), batch_size=4, num_workers=2, generator=gen, worker_init_fn=_RankWorkerSeed(7, rank))
        return [torch.cat([b for b in dl]) for _ in range(epochs)]

    r0, r1, r0_again = draws(0), draws(1), draws(0)
    assert not torch.equal(r0[0], r1[0]), 'two ranks drew the same augmentation sequence'
    assert torch.equal(r0[0], r0_again[0]) and torch.equal(r0[1], r0_again[1]), 'a rank must be reproducible'
    assert not torch.equal(r0[0], r0[1]), 'a new epoch must draw a new sequence'
