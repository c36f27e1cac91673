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


CASES = [(2, 5, 16, 16, 13, 64), (1, 3, 11, 37, 64, 64), (2, 5, 24, 32, 64, 128), (1, 1, 16, 16, 64, 64), (3, 2, 8, 8, 128, 64),
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
