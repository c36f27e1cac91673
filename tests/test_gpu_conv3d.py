"""-m gpu: the 3x3x3 convolution of the multi-date stack (BASELINE configs[3]) against torch.nn.functional.conv3d /
torch.nn.grad on the CPU.  PARITY UNPINNED against the reference: its tree holds no source for that model (UNetLSTM/ is an
empty sub-module), so torch's float32 conv3d is the only oracle there is."""
import pytest
import torch
import torch.nn.functional as F

from fabric_amd import _lib
from fabric_amd.conv3d import Conv3d3x3, to_ndhwc

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


CASES = [(2, 5, 16, 16, 13, 64), (1, 3, 11, 37, 13, 64), (3, 1, 24, 40, 13, 64), (1, 3, 11, 37, 64, 64), (2, 5, 24, 32, 64, 128), (1, 1, 16, 16, 64, 64), (3, 2, 8, 8, 128, 64),
         (1, 4, 19, 23, 192, 128)]


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', CASES)
def test_conv3d_forward_dgrad_wgrad(prec, case):
    N, D, H, W, Cin, Cout = case
    td = torch.float32 if prec == 'fp32' else torch.bfloat16
    rnd = (lambda t: t) if prec == 'fp32' else (lambda t: t.to(torch.bfloat16).float())
    x = rnd(_rand((N, Cin, D, H, W), 1))
    w = rnd(_rand((Cout, Cin, 3, 3, 3), 2, 0.1))
    b = _rand((Cout,), 3)
    dz = rnd(_rand((N, Cout, D, H, W), 4))
    ref = F.conv3d(x, w, b, padding=1)
    op = Conv3d3x3(w.cuda(), b.cuda(), precision=prec)
    xd = to_ndhwc(x.cuda(), op.cp, td)
    out, part = op.forward(xd, stats=True)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 4, 1, 2, 3)
    tol = 2e-5 if prec == 'fp32' else 1e-2
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= tol * ref.abs().max(), (got - ref).abs().max()
    # per-tile statistics (for a following BatchNorm): sum over all tiles == sum of the f32 outputs
    s = part[:, 0].double().sum(0).cpu()
    want = ref.double().sum((0, 2, 3, 4))
    assert (s - want).abs().max() <= (1e-4 if prec == 'fp32' else 2e-2) * want.abs().max() + 1e-2
    dzd = to_ndhwc(dz.cuda(), Cout, td)
    if op.wd is not None:
        dx = op.dgrad(dzd).float().cpu().permute(0, 4, 1, 2, 3)[:, :Cin]
        rdx = torch.nn.grad.conv3d_input(x.shape, w, dz, padding=1)
        assert (dx - rdx).abs().max() <= tol * rdx.abs().max()
    dw = op.wgrad(dzd, xd).cpu()
    rdw = torch.nn.grad.conv3d_weight(x, w.shape, dz, padding=1)
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert (dw - rdw).abs().max() <= (1e-4 if prec == 'fp32' else 1e-2) * rdw.abs().max()


def test_conv3d_depth_border_is_zero_padding_not_wraparound():
    """A single bright slice: its neighbours see it through exactly one depth tap, and samples do not leak into each other."""
    N, D, H, W, C = 2, 4, 8, 16, 64
    x = torch.zeros(N, C, D, H, W)
    x[0, :, D - 1] = 1.0                                   # last slice of sample 0; sample 1 starts right behind it in memory
    w = _rand((64, C, 3, 3, 3), 7, 0.05)
    op = Conv3d3x3(w.cuda(), None, precision='fp32')
    out = op.forward(to_ndhwc(x.cuda(), C, torch.float32)).cpu().permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x, w, None, padding=1)
    assert (out - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert out[1].abs().max() == 0                          # nothing of sample 0 reaches sample 1
    with pytest.raises(RuntimeError):
        _lib.call('bdn_conv3d', 2, None, 64, 0, None, 1, None, None, None, None, 1, 1, 8, 8, 64, _lib.stream_ptr())


def _torch_block(cin, cout, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    blk = nn.Sequential(nn.Conv3d(cin, cout, 3, padding=1), nn.BatchNorm3d(cout), nn.ReLU(inplace=False),
                        nn.Conv3d(cout, cout, 3, padding=1), nn.BatchNorm3d(cout), nn.ReLU(inplace=False))
    with torch.no_grad():
        for m in blk:
            if isinstance(m, nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    return blk.train()


@pytest.mark.parametrize('prec,shape', [('fp32', (2, 5, 128, 128, 13, 64)), ('bf16', (2, 5, 128, 128, 13, 64)), ('fp32', (1, 3, 21, 40, 64, 128)),
                                        ('bf16', (2, 2, 16, 48, 64, 64)), ('bf16x3', (2, 5, 128, 128, 13, 64)), ('bf16x3', (1, 3, 21, 40, 64, 128))])
def test_double_conv3d_block_matches_torch_nn(prec, shape):
    """(Conv3d + BatchNorm3d + ReLU) x 2, training mode, forward AND backward, against the same stack of stock torch.nn modules on
    the CPU -- at the multi-date benchmark shape of BASELINE configs[3] (5 dates x 13 bands x 128 x 128, two samples) and at ragged
    sizes.  Parity unpinned against the reference (it has no source for a 3-D model): torch.nn is the only oracle."""
    from fabric_amd.conv3d import DoubleConv3d
    N, D, H, W, Cin, Cout = shape
    td = torch.bfloat16 if prec == 'bf16' else torch.float32          # bf16x3: float32 tensors, split bf16 GEMM operands
    ref = _torch_block(Cin, Cout, 5)
    x = _rand((N, Cin, D, H, W), 11)
    if prec == 'bf16':
        x = x.to(torch.bfloat16).float()
    g = _rand((N, Cout, D, H, W), 12) * (_rand((N, Cout, D, H, W), 13) > 0)      # upstream gradient with exact zeros
    xr = x.clone().requires_grad_(Cin % 64 == 0)
    y = ref(xr)
    y.backward(g)
    blk = DoubleConv3d(Cin, Cout, precision=prec)
    sd0 = _torch_block(Cin, Cout, 5).state_dict()                # the block's state BEFORE the reference forward touched its buffers
    blk.load({f'conv.{k}': v for k, v in sd0.items()})
    xd = to_ndhwc(x.cuda(), blk._convs()[0].cp, td)
    out = blk.forward(xd)
    dx, grads = blk.backward(to_ndhwc(g.cuda(), Cout, td))
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 4, 1, 2, 3)
    tol = 4e-2 if prec == 'bf16' else 2e-4                       # bf16x3 is held to the float32 bound
    assert torch.isfinite(got).all()
    assert (got - y.detach()).abs().max() <= tol * y.detach().abs().max(), (got - y.detach()).abs().max()
    sd = ref.state_dict()
    for k in ('1', '4'):
        for buf in ('running_mean', 'running_var'):
            a, b = blk.P[f'conv.{k}.{buf}'].cpu(), sd[f'{k}.{buf}']
            assert (a - b).abs().max() <= (2e-2 if prec == 'bf16' else 1e-4) * max(1.0, b.abs().max().item()), (k, buf)
        assert int(blk.P[f'conv.{k}.num_batches_tracked']) == 1
    gtol = {'fp32': 2e-3, 'bf16x3': 6e-3, 'bf16': 0.25}[prec]    # relative L2 per tensor (bf16: the 2-D path's bound is 0.8; bf16x3: 6e-2 there)
    for k, p in ref.named_parameters():
        want, have = p.grad, grads[f'conv.{k}'].cpu()
        if k in ('0.bias', '3.bias'):
            assert have.abs().max() == 0 and want.abs().max() <= 1e-3 * max(1.0, g.abs().sum().item() ** 0.5)
            continue
        rel = (have - want).norm() / want.norm()
        assert rel <= gtol, (k, float(rel))
    if Cin % 64 == 0:
        rdx = xr.grad
        hdx = dx.float().cpu().permute(0, 4, 1, 2, 3)[:, :Cin]
        assert (hdx - rdx).norm() / rdx.norm() <= gtol
    else:
        assert dx is None


def test_double_conv3d_full_size_properties():
    """BASELINE configs[3] at eight samples (8 x 5 dates x 13 bands x 128 x 128), bf16: the block's forward + backward is
    bit-reproducible, finite, its BatchNorm'd output has the statistics BatchNorm promises, and it stays within the bf16
    tolerance of the exact-f32 setting of the same kernels on the same inputs."""
    from fabric_amd.conv3d import DoubleConv3d
    N, D, H, W, Cin, Cout = 8, 5, 128, 128, 13, 64
    sd0 = {f'conv.{k}': v for k, v in _torch_block(Cin, Cout, 9).state_dict().items()}
    x = _rand((N, Cin, D, H, W), 21).cuda()
    g = (_rand((N, D, H, W, Cout), 22) * 1e-3).cuda()
    res = {}
    for prec, reps in (('bf16', 2), ('fp32', 1)):
        td = torch.bfloat16 if prec == 'bf16' else torch.float32
        for r in range(reps):
            blk = DoubleConv3d(Cin, Cout, precision=prec)
            blk.load(sd0)
            out = blk.forward(to_ndhwc(x, 16, td))
            dx, grads = blk.backward(g.to(td))
            torch.cuda.synchronize()
            res[(prec, r)] = (out.float(), {k: v.clone() for k, v in grads.items()})
    o0, g0 = res[('bf16', 0)]
    o1, g1 = res[('bf16', 1)]
    assert torch.equal(o0, o1) and all(torch.equal(g0[k], g1[k]) for k in g0)          # deterministic (no float atomics)
    assert torch.isfinite(o0).all() and all(torch.isfinite(v).all() for v in g0.values())
    of, gf = res[('fp32', 0)]
    # relu(bn(.)) with gamma in [0.5, 1.5], beta in [-0.3, 0.3]: pre-ReLU channel means equal beta, so the output is non-negative
    # and a solid fraction of it is exactly zero
    assert o0.min() >= 0 and 0.2 < (o0 == 0).float().mean() < 0.8
    assert (o0 - of).abs().max() <= 4e-2 * of.abs().max()
    for k in gf:
        if gf[k].abs().max() > 0:
            assert (g0[k] - gf[k]).norm() / gf[k].norm() <= 0.25, k
