"""-m gpu parity tests, one per C-ABI entry point, against the CPU oracle (oracle/bidate_oracle.py).

Tolerances: 'fp32' = f32 storage + f32 MFMA: 2e-5 of the result's max magnitude (summation-order
noise only).  'bf16': inputs are pre-rounded to bf16 so the only differences are the bf16 rounding
of the *output* (2^-9 relative) and fp32 summation order: 1e-2 of max magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fabric_amd import _lib
from fabric_amd._lib import BDN_BF16, IN_BNRELU, IN_PLAIN
from oracle import bidate_oracle as O
from tests.gpu_util import (DT, assert_close, assert_masked, preact, bn_table, bnrelu_ref, dev, frag_to_dense, from_nhwc, pack_w, rnd, st,
                            to_nhwc)

pytestmark = pytest.mark.gpu
TOL = {'fp32': 2e-5, 'bf16': 1e-2}
PRECS = ['fp32', 'bf16']


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------ conv3x3 forward (+ stats) and BN finalize
CONV_CASES = [
    # N, H, W, C0real, C0, C1, Cout, bnrelu, ipg
    (4, 32, 32, 13, 16, 0, 64, False, 2),      # first layer: 13 bands padded to 16, two date groups
    (2, 16, 16, 64, 64, 0, 128, True, 1),      # BN+ReLU applied on load, one image per group
    (2, 24, 20, 64, 64, 64, 64, False, 2),     # two-source K loop (skip | upsampled), ragged tiles
    (4, 8, 8, 128, 128, 0, 128, True, 2),      # 8x8 maps: two images per tile
    (2, 5, 5, 64, 64, 0, 64, False, 2),        # odd tiny map (90-pixel patches end at 5x5)
    (1, 11, 45, 64, 64, 0, 192, True, 1),      # odd sizes, Cout not a multiple of 128
    (3, 17, 16, 128, 128, 128, 256, False, 3), # three images, two sources
    (8, 64, 64, 64, 64, 0, 64, True, 4),       # 256 tiles: two-stage statistics reduction
]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv3x3_forward_stats_finalize(prec, case):
    N, H, W, c0r, C0, C1, Cout, bnrelu, ipg = case
    dt, td = DT[prec]
    G = N // ipg
    x0 = rnd(prec, _rand((N, C0, H, W), 1))
    x0[:, c0r:] = 0
    x1 = rnd(prec, _rand((N, C1, H, W), 2)) if C1 else None
    w = _rand((Cout, c0r + C1, 3, 3), 3, (2.0 / (9 * (c0r + C1))) ** 0.5)
    b = _rand((Cout,), 4, 0.1)
    bn_in = bn_table(G, C0, 5) if bnrelu else None
    # ---- oracle
    a0 = bnrelu_ref(prec, x0, bn_in, ipg) if bnrelu else x0
    a = torch.cat([a0[:, :c0r], x1], 1) if C1 else a0[:, :c0r]
    z_ref = O.conv3x3(a, rnd(prec, w), b)
    # ---- device
    wp = torch.zeros(Cout, C0 + C1, 3, 3)
    wp[:, :c0r] = w[:, :c0r]
    if C1:
        wp[:, C0:] = w[:, c0r:]
    wf, _ = pack_w(prec, wp, C0 + C1)
    d0, d1 = to_nhwc(prec, x0), (to_nhwc(prec, x1) if C1 else None)
    out = torch.empty(N, H, W, Cout, dtype=td, device='cuda')
    nt = _lib.load().bdn_conv3x3_num_mtiles(N, H, W, Cout, ipg)
    stats = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    dbn = dev(bn_in) if bnrelu else None
    db = dev(b)
    _lib.call('bdn_conv3x3', dt, d0.data_ptr(), C0, d1.data_ptr() if C1 else None, C1,
              IN_BNRELU if bnrelu else IN_PLAIN, dbn.data_ptr() if bnrelu else None, ipg,
              wf.data_ptr(), db.data_ptr(), out.data_ptr(), stats.data_ptr(), N, H, W, Cout, st())
    torch.cuda.synchronize()
    assert_close('conv out', from_nhwc(out), z_ref, TOL[prec])
    # ---- stats partials -> BatchNorm table and running buffers
    gamma, beta = _rand((Cout,), 6).abs() + 0.5, _rand((Cout,), 7, 0.3)
    rm0, rv0 = _rand((Cout,), 8, 0.2), _rand((Cout,), 9).abs() + 0.5
    drm, drv = dev(rm0), dev(rv0)
    nbt = torch.zeros(1, dtype=torch.int64, device='cuda')
    bn = torch.empty(G, 4, Cout, device='cuda')
    dg, dbeta = dev(gamma), dev(beta)
    fws = torch.empty(_lib.load().bdn_bn_finalize_workspace_bytes(nt, G, Cout) // 8, dtype=torch.float64, device='cuda')
    _lib.call('bdn_bn_finalize', stats.data_ptr(), nt, G, Cout, ipg * H * W, dg.data_ptr(), dbeta.data_ptr(),
              1e-5, 0.1, drm.data_ptr(), drv.data_ptr(), nbt.data_ptr(), bn.data_ptr(), fws.data_ptr(), st())
    torch.cuda.synchronize()
    bn = bn.cpu()
    rm, rv = rm0.clone(), rv0.clone()
    tol_s = 5e-5 if prec == 'fp32' else 2e-3        # bf16: statistics come from the f32 accumulators, z_ref is exact
    for g in range(G):
        zg = z_ref[g * ipg:(g + 1) * ipg].double()
        mean = zg.mean((0, 2, 3))
        var = zg.var((0, 2, 3), unbiased=False)
        n = ipg * H * W
        inv = 1 / torch.sqrt(var + 1e-5)
        assert_close(f'mean g{g}', bn[g, 0], mean.float(), tol_s, 1e-5)
        assert_close(f'invstd g{g}', bn[g, 1], inv.float(), tol_s)
        assert_close(f'scale g{g}', bn[g, 2], (gamma * inv).float(), tol_s)
        assert_close(f'shift g{g}', bn[g, 3], (beta - mean * gamma * inv).float(), tol_s, 1e-5)
        rm = 0.9 * rm + 0.1 * mean.float()
        rv = 0.9 * rv + 0.1 * (var * n / max(n - 1, 1)).float()
    assert_close('running_mean', drm.cpu(), rm, tol_s, 1e-6)
    assert_close('running_var', drv.cpu(), rv, tol_s, 1e-6)
    assert int(nbt.item()) == G


# ------------------------------------------------------------------ data gradient through the same kernel
@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 128, 1), (2, 8, 8, 256, 128, 2), (1, 22, 45, 192, 64, 1)])
def test_conv3x3_dgrad(prec, case):
    N, H, W, Cin, Cout, ipg = case
    dt, td = DT[prec]
    dz = rnd(prec, _rand((N, Cout, H, W), 11))
    w = rnd(prec, _rand((Cout, Cin, 3, 3), 12, 0.05))
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dz, padding=1)
    _, wd = pack_w(prec, w, Cin)
    ddz = to_nhwc(prec, dz)
    out = torch.empty(N, H, W, Cin, dtype=td, device='cuda')
    _lib.call('bdn_conv3x3', dt, ddz.data_ptr(), Cout, None, 0, IN_PLAIN, None, ipg, wd.data_ptr(), None,
              out.data_ptr(), None, N, H, W, Cin, st())
    torch.cuda.synchronize()
    assert_close('dgrad', from_nhwc(out), ref, TOL[prec])


@pytest.mark.parametrize('case', [(2, 40, 24, 64, 64, 1, True), (4, 32, 32, 64, 128, 2, False), (2, 24, 20, 64, 64, 2, False),
                                  (16, 64, 64, 64, 256, 8, True), (3, 17, 45, 64, 128, 3, True)])
def test_conv3x3_dgrad_with_bn_backward_on_load(case):
    """bdn_conv3x3_dgrad_bb = bdn_bn_bwd_apply + bdn_conv3x3(_dgrad_bs) without the pass in between.  Its operand is the MASKED gradient g
    the fused producers store, and it forms dz = a g + b z + c (three per-channel constants, round 5) instead of bdn_bn_bwd_apply's
    scale (g - s0/M - xhat s1/M): the same value up to float32 rounding, so dz (the by-product) and the data gradient are held to the
    kernel tolerances against the two-kernel path, with and without the fused BatchNorm-backward statistics of the producing layer; the
    three single-chunk instantiations, ragged tiles."""
    N, H, W, C0, Cout, ipg, with_bs = case
    lib = _lib.load()
    G = N // ipg
    z = rnd('bf16', _rand((N, C0, H, W), 62))
    bn = bn_table(G, C0, 63)
    pre = preact(z, bn, ipg)
    dA = rnd('bf16', _rand((N, C0, H, W), 61)) * (pre > 1e-4)         # g: masked, and zero at the switching point whichever way it rounds
    w = rnd('bf16', _rand((C0, Cout, 3, 3), 64, 0.05))          # layer L: Cout -> C0 channels; its data gradient maps C0 -> Cout
    _, wd = pack_w('bf16', w, Cout)
    dA_d, z_d, bn_d = to_nhwc('bf16', dA), to_nhwc('bf16', z), dev(bn)
    # reference path: reduce + finalize + apply, then the plain data-gradient convolution
    ws = torch.empty(lib.bdn_bn_bwd_workspace_bytes(BDN_BF16, N, H, W, C0, ipg) // 4, device='cuda')
    sums = torch.empty(G, 2, C0, device='cuda')
    dg, db = torch.empty(C0, device='cuda'), torch.empty(C0, device='cuda')
    dz_ref = torch.empty(N, H, W, C0, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_bn_bwd', BDN_BF16, dA_d.data_ptr(), C0, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, C0,
              ws.data_ptr(), sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dz_ref.data_ptr(), st())
    out_ref = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device='cuda')
    nt = lib.bdn_conv3x3_num_mtiles(N, H, W, Cout, ipg)
    zp = to_nhwc('bf16', rnd('bf16', _rand((N, Cout, H, W), 65)))
    bnp = dev(bn_table(G, Cout, 66))
    part_ref = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    part = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    if with_bs:
        _lib.call('bdn_conv3x3_dgrad_bs', BDN_BF16, dz_ref.data_ptr(), C0, wd.data_ptr(), out_ref.data_ptr(), zp.data_ptr(), bnp.data_ptr(), ipg,
                  part_ref.data_ptr(), N, H, W, Cout, st())
    else:
        _lib.call('bdn_conv3x3', BDN_BF16, dz_ref.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg, wd.data_ptr(), None, out_ref.data_ptr(), None,
                  N, H, W, Cout, st())
    out = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    dz = torch.full((N, H, W, C0), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_conv3x3_dgrad_bb', BDN_BF16, dA_d.data_ptr(), C0, z_d.data_ptr(), bn_d.data_ptr(), sums.data_ptr(), ipg, wd.data_ptr(),
              out.data_ptr(), zp.data_ptr() if with_bs else None, bnp.data_ptr() if with_bs else None, part.data_ptr() if with_bs else None,
              dz.data_ptr(), N, H, W, Cout, st())
    torch.cuda.synchronize()
    # dz: two roundings of nearly the same float32 value -- equal, or one bf16 step apart
    assert_close('dz (a g + b z + c)', dz.float().cpu(), dz_ref.float().cpu(), 8e-3)
    assert (dz.float() - dz_ref.float()).abs().max() <= 2.0 ** -7 * dz_ref.float().abs().max()
    same = (dz == dz_ref).float().mean().item()
    assert same > 0.9, f'only {same:.3f} of the dz values are bit-identical to bn_bwd_apply'
    # the data gradient of two operands that differ by single bf16 steps in a few percent of the entries
    live = out_ref.float() != 0 if with_bs else torch.ones_like(out_ref, dtype=torch.bool)       # the fused statistics store the masked gradient
    assert_close('dA_prev', out.float().cpu(), out_ref.float().cpu(), 8e-3)
    if with_bs:
        assert torch.equal(out == 0, out_ref == 0) or ((out == 0) != (out_ref == 0)).float().mean() < 1e-3
        assert_close('fused statistics', part.cpu(), part_ref.cpu(), 2e-2, abs_floor=1e-3)
    # without the by-product store the data gradient is the same, bit for bit
    out2 = torch.empty_like(out)
    _lib.call('bdn_conv3x3_dgrad_bb', BDN_BF16, dA_d.data_ptr(), C0, z_d.data_ptr(), bn_d.data_ptr(), sums.data_ptr(), ipg, wd.data_ptr(),
              out2.data_ptr(), None, None, None, None, N, H, W, Cout, st())
    torch.cuda.synchronize()
    if with_bs:
        assert torch.equal(out2[out != 0], out[out != 0])       # the unmasked launch stores dA where the masked one stores g = dA
    else:
        assert torch.equal(out2, out)


# ------------------------------------------------------------------ weight gradient
WG_CASES = [
    # N, H, W, C0real, C0, C1, Cout, bnrelu, ipg
    (4, 32, 32, 13, 16, 0, 64, False, 2),
    (2, 16, 16, 64, 64, 0, 128, True, 1),
    (2, 24, 20, 64, 64, 64, 64, False, 2),
    (4, 8, 8, 128, 128, 0, 64, True, 2),
    (2, 5, 11, 64, 64, 0, 64, False, 1),
    (8, 16, 16, 64, 64, 0, 64, True, 4),
]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', WG_CASES)
def test_conv3x3_wgrad(prec, case):
    N, H, W, c0r, C0, C1, Cout, bnrelu, ipg = case
    dt, td = DT[prec]
    G = N // ipg
    x0 = rnd(prec, _rand((N, C0, H, W), 21))
    x0[:, c0r:] = 0
    x1 = rnd(prec, _rand((N, C1, H, W), 22)) if C1 else None
    dz = rnd(prec, _rand((N, Cout, H, W), 23))
    bn_in = bn_table(G, C0, 24) if bnrelu else None
    a0 = bnrelu_ref(prec, x0, bn_in, ipg) if bnrelu else x0
    a = torch.cat([a0[:, :c0r], x1], 1) if C1 else a0[:, :c0r]
    ref = torch.nn.grad.conv2d_weight(a.double(), (Cout, c0r + C1, 3, 3), dz.double(), padding=1).float()
    d0, d1, ddz = to_nhwc(prec, x0), (to_nhwc(prec, x1) if C1 else None), to_nhwc(prec, dz)
    nbytes = _lib.load().bdn_wgrad_workspace_bytes(N, H, W, Cout, C0 + C1, ipg)
    part = torch.empty(nbytes // 4, device='cuda')
    cin_real = c0r + C1 if not C1 else C0 + C1
    dw = torch.full((Cout, cin_real, 3, 3), float('nan'), device='cuda')
    dbn = dev(bn_in) if bnrelu else None
    _lib.call('bdn_conv3x3_wgrad', dt, ddz.data_ptr(), Cout, d0.data_ptr(), C0, d1.data_ptr() if C1 else None, C1,
              IN_BNRELU if bnrelu else IN_PLAIN, dbn.data_ptr() if bnrelu else None, ipg,
              part.data_ptr(), dw.data_ptr(), cin_real, N, H, W, st())
    torch.cuda.synchronize()
    got = dw.cpu()
    if C1:   # the padded first source keeps its zero channels in this layout
        got = torch.cat([got[:, :c0r], got[:, C0:]], 1)
    assert_close('wgrad', got, ref, 2e-5 if prec == 'fp32' else 2e-5)   # inputs pre-rounded, f32 accumulate & output


# ------------------------------------------------------------------ BatchNorm + ReLU backward
@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', [(4, 16, 16, 64, 2, 0), (2, 9, 7, 128, 1, 64), (6, 8, 8, 512, 3, 0)])
def test_bn_bwd(prec, case):
    N, H, W, C, ipg, extra = case
    dt, td = DT[prec]
    G = N // ipg
    ld = C + extra
    z = rnd(prec, _rand((N, C, H, W), 31))
    dA_full = rnd(prec, _rand((N, ld, H, W), 32))
    gamma = _rand((C,), 33).abs() + 0.5
    gamma[::7] *= -1
    beta = _rand((C,), 34, 0.3)
    # table from the true batch statistics of z
    bn = torch.empty(G, 4, C)
    zs = z.clone().double().requires_grad_(True)
    ys = []
    for g in range(G):
        zg = zs[g * ipg:(g + 1) * ipg]
        mean, var = zg.mean((0, 2, 3)), zg.var((0, 2, 3), unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        bn[g, 0], bn[g, 1] = mean.detach().float(), inv.detach().float()
        bn[g, 2] = (gamma * inv.detach()).float()
        bn[g, 3] = (beta - mean.detach() * gamma * inv.detach()).float()
        y = (zg - mean[None, :, None, None]) * (inv * gamma.double())[None, :, None, None] + beta.double()[None, :, None, None]
        ys.append(torch.relu(y))
    gd = dA_full[:, extra:extra + C].double()
    gs = gamma.double().clone().requires_grad_(True)     # for dgamma use an explicit formula below
    torch.cat(ys).backward(gd)
    dz_ref = zs.grad.float()
    # dgamma / dbeta by the explicit formula
    a = torch.cat(ys).detach()
    gm = gd * (a > 0)
    dbeta_ref = gm.sum((0, 2, 3)).float()
    xhat = torch.cat([(z[g * ipg:(g + 1) * ipg].double() - bn[g, 0].double()[None, :, None, None]) * bn[g, 1].double()[None, :, None, None] for g in range(G)])
    dgamma_ref = (gm * xhat).sum((0, 2, 3)).float()
    # ---- device
    dz_d = torch.empty(N, H, W, C, dtype=td, device='cuda')
    dA_d = to_nhwc(prec, dA_full)
    z_d = to_nhwc(prec, z)
    wsb = torch.empty(_lib.load().bdn_bn_bwd_workspace_bytes(dt, N, H, W, C, ipg) // 4, device='cuda')
    sums = torch.empty(G, 2, C, device='cuda')
    dgam, dbet = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    bn_d = dev(bn)
    es = 2 if prec == 'bf16' else 4
    _lib.call('bdn_bn_bwd', dt, dA_d.data_ptr() + extra * es, ld, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, C,
              wsb.data_ptr(), sums.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), dz_d.data_ptr(), st())
    torch.cuda.synchronize()
    assert_close('dbeta', dbet.cpu(), dbeta_ref, 1e-4)
    assert_close('dgamma', dgam.cpu(), dgamma_ref, 1e-4)
    assert_close('dz', from_nhwc(dz_d), dz_ref, 1e-4 if prec == 'fp32' else 1e-2)


# ------------------------------------------------------------------ pool / product / upsample and their backward
@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(4, 16, 16, 64, 2), (2, 45, 22, 64, 1), (2, 11, 11, 128, 1)])
def test_bnrelu_pool(prec, shape):
    N, H, W, C, ipg = shape
    dt, td = DT[prec]
    z = rnd(prec, _rand((N, C, H, W), 41))
    bn = bn_table(N // ipg, C, 42)
    ref = O.maxpool2(bnrelu_ref(prec, z, bn, ipg))
    out = torch.empty(N, H // 2, W // 2, C, dtype=td, device='cuda')
    z_d, bn_d = to_nhwc(prec, z), dev(bn)
    _lib.call('bdn_bnrelu_pool', dt, z_d.data_ptr(), bn_d.data_ptr(), ipg, out.data_ptr(), N, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('pool', from_nhwc(out), ref, 1e-6 if prec == 'fp32' else 8e-3)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(4, 16, 16, 64, 2), (2, 45, 22, 64, 1), (3, 7, 5, 512, 3), (2, 9, 9, 16, 1)])
def test_bnrelu_materialised(prec, shape):
    """bdn_bnrelu writes exactly the tensor the 3x3 consumers derive on load (models/unet_parts.py:14-15)."""
    N, H, W, C, ipg = shape
    dt, td = DT[prec]
    z = rnd(prec, _rand((N, C, H, W), 141))
    bn = bn_table(N // ipg, C, 142)
    ref = bnrelu_ref(prec, z, bn, ipg)
    out = torch.full((N, H, W, C), float('nan'), dtype=td, device='cuda')
    z_d, bn_d = to_nhwc(prec, z), dev(bn)
    _lib.call('bdn_bnrelu', dt, z_d.data_ptr(), bn_d.data_ptr(), ipg, out.data_ptr(), N, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('bnrelu', from_nhwc(out), ref, 1e-6 if prec == 'fp32' else 8e-3)
    with pytest.raises(RuntimeError):
        _lib.call('bdn_bnrelu', dt, z_d.data_ptr(), bn_d.data_ptr(), ipg, out.data_ptr(), N, H, W, C + 1, st())


@pytest.mark.parametrize('prec', PRECS)
def test_fuse_product(prec):
    B, H, W, C = 3, 10, 12, 64
    dt, td = DT[prec]
    z = rnd(prec, _rand((2 * B, C, H, W), 43))
    bn = bn_table(2, C, 44)
    a = bnrelu_ref(prec, z, bn, B)
    ref = torch.relu(a[B:] * a[:B])                      # models/bidate_model.py:35
    out = torch.empty(B, H, W, C, dtype=td, device='cuda')
    z_d, bn_d = to_nhwc(prec, z), dev(bn)
    _lib.call('bdn_fuse_product', dt, z_d.data_ptr(), bn_d.data_ptr(), out.data_ptr(), B, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('product', from_nhwc(out), ref, 1e-6 if prec == 'fp32' else 8e-3)


UP_CASES = [(2, 8, 8, 16, 16, 64), (2, 5, 5, 11, 11, 64), (1, 22, 22, 45, 45, 32), (2, 1, 1, 2, 3, 16), (1, 3, 7, 6, 14, 64),
            (2, 22, 19, 45, 39, 128), (1, 16, 32, 32, 64, 256), (1, 9, 8, 18, 17, 64)]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', UP_CASES)
@pytest.mark.parametrize('bnrelu', [False, True])
def test_upsample2x_and_backward(prec, case, bnrelu):
    B, h, w, H, W, C = case
    dt, td = DT[prec]
    src = rnd(prec, _rand((B, C, h, w), 45))
    bn = bn_table(1, C, 46)
    a = bnrelu_ref(prec, src, bn, B) if bnrelu else src
    skip = torch.zeros(B, 1, H, W)
    a_req = a.clone().double().requires_grad_(True)
    ref = O.pad_to(O.upsample2x_align(a_req), skip)
    # cross-check the oracle's bilinear against torch's own
    tref = O.pad_to(F.interpolate(a.double(), scale_factor=2, mode='bilinear', align_corners=True), skip)
    assert (ref.detach() - tref).abs().max() < 1e-6
    out = torch.empty(B, H, W, C, dtype=td, device='cuda')
    s_d, bn_d = to_nhwc(prec, src), dev(bn)
    _lib.call('bdn_upsample2x', dt, s_d.data_ptr(), IN_BNRELU if bnrelu else IN_PLAIN, bn_d.data_ptr(),
              out.data_ptr(), B, h, w, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('upsample', from_nhwc(out), ref.detach().float(), 1e-5 if prec == 'fp32' else 8e-3)
    if bnrelu:
        return
    # backward: dU lives in a wider tensor (channel slice), like the decoder's [dF | dU] gradient
    extra = 16
    dU = rnd(prec, _rand((B, C + extra, H, W), 47))
    ref.backward(dU[:, extra:].double())
    dsrc = torch.empty(B, h, w, C, dtype=td, device='cuda')
    dU_d = to_nhwc(prec, dU)
    es = 2 if prec == 'bf16' else 4
    _lib.call('bdn_upsample2x_bwd', dt, dU_d.data_ptr() + extra * es, C + extra, dsrc.data_ptr(), B, h, w, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('upsample bwd', from_nhwc(dsrc), a_req.grad.float(), 1e-5 if prec == 'fp32' else 8e-3)
    # the same pass with the BatchNorm-backward partial sums of the upsampled layer fused in (where the tiled kernel takes the shape):
    # identical dsrc, and sum g / sum g*z over the STORED gradient, g = dsrc * [scale*z + shift > 0]
    rows = _lib.load().bdn_upsample2x_bwd_rows(dt, B, h, w, C)
    if rows == 0:
        with pytest.raises(RuntimeError, match='outside the tiled kernel'):
            t = torch.zeros(16, device='cuda')
            _lib.call('bdn_upsample2x_bwd_bs', dt, dU_d.data_ptr() + extra * es, C + extra, dsrc.data_ptr(), t.data_ptr(), t.data_ptr(),
                      t.data_ptr(), B, h, w, H, W, C, st())
        return
    zp = rnd(prec, _rand((B, C, h, w), 48))
    bnp = bn_table(1, C, 49)
    dsrc2 = torch.empty_like(dsrc)
    part = torch.full((rows, 2, C), float('nan'), device='cuda')
    zp_d, bnp_d = to_nhwc(prec, zp), dev(bnp)
    _lib.call('bdn_upsample2x_bwd_bs', dt, dU_d.data_ptr() + extra * es, C + extra, dsrc2.data_ptr(), zp_d.data_ptr(), bnp_d.data_ptr(),
              part.data_ptr(), B, h, w, H, W, C, st())
    torch.cuda.synchronize()
    assert_masked('upsample bwd', from_nhwc(dsrc2), from_nhwc(dsrc), preact(zp, bnp, B))      # what is stored is g, the masked gradient
    g = from_nhwc(dsrc2).double()
    got = part.cpu().double().sum(0)
    assert_close('upsample bwd: sum g', got[0], g.sum((0, 2, 3)), 2e-5 if prec == 'fp32' else 1e-4, 1e-4)
    assert_close('upsample bwd: sum g z', got[1], (g * zp.double()).sum((0, 2, 3)), 2e-5 if prec == 'fp32' else 1e-4, 1e-4)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, True), (2, 11, 45, 64, True), (3, 8, 8, 128, False), (1, 5, 5, 64, True)])
def test_enc_skip_bwd(prec, case):
    B, H, W, C, pooled = case
    dt, td = DT[prec]
    z = rnd(prec, _rand((2 * B, C, H, W), 51))
    bn = bn_table(2, C, 52)
    extra = 32
    dF = rnd(prec, _rand((B, C + extra, H, W), 53))
    dP = rnd(prec, _rand((2 * B, C, H // 2, W // 2), 54)) if pooled else None
    a = bnrelu_ref(prec, z, bn, B).double().requires_grad_(True)
    loss = (torch.relu(a[B:] * a[:B]) * dF[:, :C].double()).sum()
    if pooled:
        loss = loss + (O.maxpool2(a) * dP.double()).sum()
    loss.backward()
    ref = a.grad.float()
    out = torch.empty(2 * B, H, W, C, dtype=td, device='cuda')
    dF_d, z_d, bn_d = to_nhwc(prec, dF), to_nhwc(prec, z), dev(bn)
    dP_d = to_nhwc(prec, dP) if pooled else None
    rows = _lib.load().bdn_enc_skip_bwd_rows(dt, B, H, W, C)
    part = torch.full((2, rows, 2, C), float('nan'), device='cuda')
    _lib.call('bdn_enc_skip_bwd', dt, dF_d.data_ptr(), C + extra, z_d.data_ptr(), bn_d.data_ptr(),
              dP_d.data_ptr() if pooled else None, out.data_ptr(), part.data_ptr(), B, H, W, C, st())
    out2 = torch.empty_like(out)
    _lib.call('bdn_enc_skip_bwd', dt, dF_d.data_ptr(), C + extra, z_d.data_ptr(), bn_d.data_ptr(),
              dP_d.data_ptr() if pooled else None, out2.data_ptr(), None, B, H, W, C, st())
    torch.cuda.synchronize()
    assert_masked('enc_skip_bwd', from_nhwc(out), from_nhwc(out2), preact(z, bn, B))           # with the statistics the stored gradient is g (masked)
    # fused BatchNorm-backward partial sums: sum g and sum g*z over the STORED gradient, per date
    g = from_nhwc(out).double()
    for d in range(2):
        sl = slice(d * B, (d + 1) * B)
        got = part[d].double().sum(0).cpu()                       # [2][C]
        assert_close(f'sum g (date {d})', got[0].float(), g[sl].sum((0, 2, 3)).float(), 2e-5, abs_floor=1e-4)
        assert_close(f'sum g*z (date {d})', got[1].float(), (g[sl] * z[sl].double()).sum((0, 2, 3)).float(), 2e-5, abs_floor=1e-4)
    # Where a == 0 the reference's relu(a2*a1) has zero slope while the kernel returns dF*a_other; both are
    # multiplied by the producer's own ReLU mask [a > 0] in the very next step (bn_bwd), so compare there.
    live = (a.detach() > 0).float()
    assert_close('enc_skip_bwd', from_nhwc(out) * live, ref * live, 1e-6 if prec == 'fp32' else 8e-3)


# ------------------------------------------------------------------ classifier
@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(2, 16, 16, 64, 2), (1, 9, 13, 64, 3)])
def test_outc_fwd_bwd(prec, shape):
    B, H, W, C, ncls = shape
    dt, td = DT[prec]
    z = rnd(prec, _rand((B, C, H, W), 61))
    bn = bn_table(1, C, 62)
    w, b = _rand((ncls, C, 1, 1), 63, 0.2), _rand((ncls,), 64, 0.1)
    a = bnrelu_ref(prec, z, bn, B).double().requires_grad_(True)
    wd_, bd_ = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = O.conv1x1(a, wd_, bd_)
    logits = torch.empty(B, ncls, H, W, device='cuda')
    z_d, bn_d, w_d, b_d = to_nhwc(prec, z), dev(bn), dev(w), dev(b)
    _lib.call('bdn_outc_fwd', dt, z_d.data_ptr(), bn_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr(), logits.data_ptr(), B, H, W, C, ncls, st())
    torch.cuda.synchronize()
    assert_close('logits', logits.cpu(), ref.detach().float(), 1e-5)
    dl = _rand((B, ncls, H, W), 65)
    ref.backward(dl.double())
    dA = torch.empty(B, H, W, C, dtype=td, device='cuda')
    dw, db = torch.empty(ncls, C, device='cuda'), torch.empty(ncls, device='cuda')
    dl_d = dev(dl)
    rows = _lib.load().bdn_outc_bwd_rows(dt, B, H, W, C)
    part = torch.full((rows, 2, C), float('nan'), device='cuda')
    wsz = _lib.load().bdn_outc_bwd_workspace_bytes(dt, B, H, W, C, ncls) // 4
    ows = torch.full((wsz,), float('nan'), device='cuda')
    _lib.call('bdn_outc_bwd', dt, dl_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), w_d.data_ptr(), dA.data_ptr(),
              dw.data_ptr(), db.data_ptr(), part.data_ptr(), ows.data_ptr(), B, H, W, C, ncls, st())
    dA2 = torch.empty_like(dA)
    dw2, db2 = torch.full_like(dw, 5.0), torch.full_like(db, 5.0)
    _lib.call('bdn_outc_bwd', dt, dl_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), w_d.data_ptr(), dA2.data_ptr(),
              dw2.data_ptr(), db2.data_ptr(), None, ows.data_ptr(), B, H, W, C, ncls, st())
    torch.cuda.synchronize()
    assert torch.equal(dA, dA2) and torch.equal(dw, dw2) and torch.equal(db, db2)       # fixed-order sums: same bits every time
    gm = from_nhwc(dA).double() * (bnrelu_ref(prec, z, bn, B) > 0)          # fused BatchNorm-backward partial sums
    got = part.double().sum(0).cpu()
    assert_close('sum g', got[0].float(), gm.sum((0, 2, 3)).float(), 2e-5, abs_floor=1e-4)
    assert_close('sum g*z', got[1].float(), (gm * z.double()).sum((0, 2, 3)).float(), 2e-5, abs_floor=1e-4)
    assert_close('dA', from_nhwc(dA), a.grad.float(), 1e-5 if prec == 'fp32' else 8e-3)
    assert_close('dw', dw.cpu(), wd_.grad.float().reshape(ncls, C), 1e-4)
    assert_close('db', db.cpu(), bd_.grad.float(), 1e-4)


# ------------------------------------------------------------------ Tversky loss
@pytest.mark.parametrize('shape', [(3, 2, 24, 20), (2, 3, 16, 300), (64, 2, 128, 128)])
def test_tversky(shape):
    B, ncls, H, W = shape
    logits = _rand((B, ncls, H, W), 71)
    labels = torch.from_numpy((np.random.default_rng(72).uniform(0, 1, (B, H, W)) < 0.2).astype(np.uint8))
    lg = logits.double().requires_grad_(True)
    ref = O.tversky_loss(lg, labels.long(), 0.1, 0.9)
    ref.backward()
    ws = torch.empty(_lib.load().bdn_overlap_workspace_bytes(B, ncls, H, W, 0) // 4, device='cuda')
    loss = torch.empty(1, device='cuda')
    counts = torch.empty(4, dtype=torch.int32, device='cuda')
    dl = torch.empty(B, ncls, H, W, device='cuda')
    lg_d, lb_d = dev(logits), labels.cuda()
    _lib.call('bdn_tversky', lg_d.data_ptr(), lb_d.data_ptr(), 0.1, 0.9, 1e-7, ws.data_ptr(), loss.data_ptr(),
              counts.data_ptr(), dl.data_ptr(), B, ncls, H, W, st())
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 2e-6
    assert_close('dlogits', dl.cpu(), lg.grad.float(), 2e-4)
    preds = logits.argmax(1)
    lab = labels.long()
    exp = [int(((preds == 1) & (lab == 1)).sum()), int(((preds == 1) & (lab != 1)).sum()),
           int(((preds != 1) & (lab == 1)).sum()), int((preds == lab).sum())]
    assert counts.cpu().tolist() == exp


def test_tversky_golden(golden_dir):
    """utils/metrics.py:130-171 value + gradient captured from the reference itself (G5)."""
    import os
    g = np.load(os.path.join(golden_dir, 'g5_losses.npz'))
    logits, labels = torch.from_numpy(g['logits']), torch.from_numpy(g['labels'])
    B, ncls, H, W = logits.shape
    ws = torch.empty(_lib.load().bdn_overlap_workspace_bytes(B, ncls, H, W, 0) // 4, device='cuda')
    loss = torch.empty(1, device='cuda')
    dl = torch.empty(B, ncls, H, W, device='cuda')
    lg_d, lb_d = dev(logits), labels.cuda()
    _lib.call('bdn_tversky', lg_d.data_ptr(), lb_d.data_ptr(), 0.1, 0.9, 1e-7, ws.data_ptr(), loss.data_ptr(),
              None, dl.data_ptr(), B, ncls, H, W, st())
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g['tversky_0.1_0.9_r3'])) < 2e-6
    assert_close('dlogits', dl.cpu(), torch.from_numpy(g['dlogits_tversky_r3']), 2e-4)


# ------------------------------------------------------------------ converters, SGD
@pytest.mark.parametrize('prec', PRECS)
def test_pack_input_and_weights(prec):
    dt, td = DT[prec]
    B, C, H, W, Cp = 2, 13, 9, 7, 16
    x1, x2 = _rand((B, C, H, W), 81), _rand((B, C, H, W), 82)
    out = torch.empty(2 * B, H, W, Cp, dtype=td, device='cuda')
    a, b = dev(x1), dev(x2)
    _lib.call('bdn_pack_input', dt, a.data_ptr(), b.data_ptr(), out.data_ptr(), B, C, H, W, Cp, st())
    torch.cuda.synchronize()
    got = from_nhwc(out)
    ref = rnd(prec, torch.cat([x1, x2]))
    assert torch.equal(got[:, :C], ref) and (got[:, C:] == 0).all()
    Cp2 = 32                                      # the data-gradient image needs Cin_pad % 32 == 0
    w = _rand((64, C, 3, 3), 83)
    wf, wd = pack_w(prec, w, Cp2)
    torch.cuda.synchronize()
    wr = rnd(prec, w)
    dense_f = frag_to_dense(prec, wf, 64, Cp2)                      # [co][tap][ci]
    assert torch.equal(dense_f[:, :, :C], wr.permute(0, 2, 3, 1).reshape(64, 9, C))
    assert (dense_f[:, :, C:] == 0).all()
    dense_d = frag_to_dense(prec, wd, Cp2, 64)                      # [ci][tap][co], taps rotated by 180 degrees
    rot = torch.flip(wr, (2, 3)).permute(1, 2, 3, 0).reshape(C, 9, 64)
    assert torch.equal(dense_d[:C], rot) and (dense_d[C:] == 0).all()


def test_sgd_step():
    n = 1000003
    p, g = _rand((n,), 91), _rand((n,), 92)
    pd, gd = dev(p), dev(g)
    _lib.call('bdn_sgd_step', pd.data_ptr(), gd.data_ptr(), 1e-3, 0.5, n, st())
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), p - 1e-3 * 0.5 * g, atol=1e-6, rtol=0)


# ------------------------------------------------------------------ fused BatchNorm-backward statistics
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [(4, 16, 16, 128, 64, 2), (6, 24, 40, 64, 128, 3), (4, 8, 8, 64, 64, 2), (2, 20, 18, 256, 128, 2),
                                  (2, 33, 17, 64, 64, 1), (8, 128, 128, 64, 64, 4)])
def test_dgrad_with_fused_bn_bwd_stats(prec, case):
    """bdn_conv3x3_dgrad_bs + bdn_bn_bwd_apply == bdn_conv3x3 (data gradient) + bdn_bn_bwd: the stored dA is the plain data gradient
    under the ReLU mask of the producing layer (identical where the mask is on), and dgamma / dbeta / dz equal up to the summation order
    of the partial sums (BatchNorm backward applies the same mask again)."""
    N, H, W, Cz, Cout, ipg = case           # dz has Cz channels; the data gradient has Cout channels (= producing layer's width)
    dt, td = DT[prec]
    G = N // ipg
    lib = _lib.load()
    dzin = to_nhwc(prec, rnd(prec, _rand((N, Cz, H, W), 51)))
    wrot = rnd(prec, _rand((Cout, Cz, 3, 3), 52, 0.05))          # any filter: the test is about the epilogue
    wf, _ = pack_w(prec, wrot, Cz)
    zprev = rnd(prec, _rand((N, Cout, H, W), 53))
    z_d = to_nhwc(prec, zprev)
    bn = bn_table(G, Cout, 7)
    for g in range(G):                                           # statistics of zprev itself so masks are mixed
        zg = zprev[g * ipg:(g + 1) * ipg].double()
        mean, var = zg.mean((0, 2, 3)), zg.var((0, 2, 3), unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        gamma = bn[g, 2].double() / bn[g, 1].double()
        bn[g, 0], bn[g, 1] = mean.float(), inv.float()
        bn[g, 2] = (gamma * inv).float()
        bn[g, 3] = (0.1 - mean * gamma * inv).float()
    bn_d = dev(bn)
    # reference path: plain data gradient, then the three-kernel BatchNorm backward
    dA_ref = torch.empty(N, H, W, Cout, dtype=td, device='cuda')
    _lib.call('bdn_conv3x3', dt, dzin.data_ptr(), Cz, None, 0, 0, None, ipg, wf.data_ptr(), None, dA_ref.data_ptr(), None,
              N, H, W, Cout, st())
    wsb = torch.empty(lib.bdn_bn_bwd_workspace_bytes(dt, N, H, W, Cout, ipg) // 4, device='cuda')
    sums_r = torch.empty(G, 2, Cout, device='cuda')
    dg_r, db_r = torch.empty(Cout, device='cuda'), torch.empty(Cout, device='cuda')
    dz_r = torch.empty(N, H, W, Cout, dtype=td, device='cuda')
    _lib.call('bdn_bn_bwd', dt, dA_ref.data_ptr(), Cout, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, Cout,
              wsb.data_ptr(), sums_r.data_ptr(), dg_r.data_ptr(), db_r.data_ptr(), dz_r.data_ptr(), st())
    # fused path
    nt = lib.bdn_conv3x3_num_mtiles(N, H, W, Cout, ipg)
    assert nt % G == 0
    part = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    dA = torch.empty_like(dA_ref)
    _lib.call('bdn_conv3x3_dgrad_bs', dt, dzin.data_ptr(), Cz, wf.data_ptr(), dA.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(),
              ipg, part.data_ptr(), N, H, W, Cout, st())
    sums = torch.empty(G, 2, Cout, device='cuda')
    dg, db = torch.empty(Cout, device='cuda'), torch.empty(Cout, device='cuda')
    dz = torch.empty_like(dz_r)
    scratch = torch.empty(lib.bdn_bn_bwd_scratch_bytes(G, Cout), dtype=torch.uint8, device='cuda') if N * H * W > 1500 else None
    _lib.call('bdn_bn_bwd_apply', dt, dA.data_ptr(), Cout, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, Cout,
              part.data_ptr(), nt // G, 1, sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dz.data_ptr(),
              scratch.data_ptr() if scratch is not None else None, st())
    torch.cuda.synchronize()
    assert_masked('dgrad_bs', from_nhwc(dA), from_nhwc(dA_ref), preact(zprev, bn, ipg))       # identical where the ReLU is on, zero where it is off
    assert torch.isfinite(part).all()
    assert_close('dbeta', db.cpu(), db_r.cpu(), 2e-5)
    assert_close('dgamma', dg.cpu(), dg_r.cpu(), 2e-5, abs_floor=2e-4)
    assert_close('dz', dz.float().cpu(), dz_r.float().cpu(), 8e-3 if prec == 'bf16' else 2e-5)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(2, 16, 16, 64, 2, 1), (4, 33, 20, 64, 2, 2), (2, 24, 24, 128, 3, 2), (2, 128, 128, 64, 2, 2)])
def test_head_bn_bwd_recomputed_from_dlogits(prec, shape):
    """bdn_outc_bwd(dA = NULL) + bdn_bn_bwd_finalize + bdn_outc_bn_bwd_apply == bdn_outc_bwd + bdn_bn_bwd_apply, bit for bit
    (dz, sums, dgamma, dbeta): the classifier's data gradient is re-formed and re-rounded exactly as outc_bwd stores it."""
    B, H, W, C, ncls, ipg = shape
    dt, td = DT[prec]
    G = B // ipg
    lib = _lib.load()
    z = rnd(prec, _rand((B, C, H, W), 501))
    z_d = to_nhwc(prec, z)
    bn = bn_table(G, C, 502)
    bn1 = bn[:1].clone()                                   # outc_bwd works on one statistic group (the decoder's)
    bn_use = bn if G > 1 else bn1
    w_d = dev(_rand((ncls, C), 503, 0.2))
    dl_d = dev(_rand((B, ncls, H, W), 504))
    rows = lib.bdn_outc_bwd_rows(dt, B, H, W, C)
    out = {}
    for fused in (0, 1):
        dA = torch.empty(B, H, W, C, dtype=td, device='cuda')
        dw, db = torch.empty(ncls, C, device='cuda'), torch.empty(ncls, device='cuda')
        part = torch.full((rows, 2, C), float('nan'), device='cuda')
        bn_d = dev(bn1)
        ows = torch.empty(lib.bdn_outc_bwd_workspace_bytes(dt, B, H, W, C, ncls) // 4, device='cuda')
        _lib.call('bdn_outc_bwd', dt, dl_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), w_d.data_ptr(),
                  None if fused else dA.data_ptr(), dw.data_ptr(), db.data_ptr(), part.data_ptr(), ows.data_ptr(), B, H, W, C, ncls, st())
        sums = torch.empty(1, 2, C, device='cuda')
        dg, dbt = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        dz = torch.full((B, H, W, C), float('nan'), dtype=td, device='cuda')
        if fused:
            _lib.call('bdn_bn_bwd_finalize', bn_d.data_ptr(), 1, C, part.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(),
                      dbt.data_ptr(), None, st())
            _lib.call('bdn_outc_bn_bwd_apply', dt, dl_d.data_ptr(), w_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), B, sums.data_ptr(),
                      dz.data_ptr(), B, H, W, C, ncls, st())
        else:
            _lib.call('bdn_bn_bwd_apply', dt, dA.data_ptr(), C, z_d.data_ptr(), bn_d.data_ptr(), B, B, H, W, C, part.data_ptr(), rows, 1,
                      sums.data_ptr(), dg.data_ptr(), dbt.data_ptr(), dz.data_ptr(), None, st())
        torch.cuda.synchronize()
        out[fused] = (dz.float().cpu(), sums.cpu(), dg.cpu(), dbt.cpu())
    assert torch.isfinite(out[1][0]).all()
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):                      # nobody would consume the gradient
        _lib.call('bdn_outc_bwd', dt, dl_d.data_ptr(), z_d.data_ptr(), dev(bn1).data_ptr(), w_d.data_ptr(), None, dw.data_ptr(),
                  db.data_ptr(), None, ows.data_ptr(), B, H, W, C, ncls, st())


@pytest.mark.parametrize('prec', PRECS)
def test_pack_weights_multi_equals_per_layer_pack(prec):
    """bdn_pack_weights_multi (one call for a list of layers; bf16 takes the LDS-tiled kernel for regular layers and the
    element-wise one for the rest) writes the same fragment-order images as bdn_pack_weights layer by layer."""
    import struct
    dt, td = DT[prec]
    layers = [(64, 13, 16, False), (64, 64, 64, True), (128, 64, 64, True), (256, 384, 384, True), (64, 128, 128, True),
              (96, 40, 48, True), (32, 64, 64, False)]                  # Cout, Cin, Cin_pad, with data-gradient image
    ws, outs, rec = [], [], b''
    for i, (co, ci, cip, has_wd) in enumerate(layers):
        if has_wd and cip % 32:
            continue
        w = dev(_rand((co, ci, 3, 3), 400 + i))
        wf = torch.full((co * 9 * cip,), 7.0, dtype=td, device='cuda')
        wd = torch.full((co * 9 * cip,), 7.0, dtype=td, device='cuda') if has_wd else None
        ws.append((w, co, ci, cip)); outs.append((wf, wd))
        rec += struct.pack('<QQQiiii', w.data_ptr(), wf.data_ptr(), wd.data_ptr() if has_wd else 0, co, ci, cip, 0)
    desc = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    _lib.call('bdn_pack_weights_multi', dt, desc.data_ptr(), len(ws), st())
    torch.cuda.synchronize()
    for (w, co, ci, cip), (wf, wd) in zip(ws, outs):
        rf = torch.empty_like(wf)
        rd = torch.empty_like(wd) if wd is not None else None
        _lib.call('bdn_pack_weights', dt, w.data_ptr(), rf.data_ptr(), rd.data_ptr() if rd is not None else None, co, ci, cip, st())
        torch.cuda.synchronize()
        assert torch.equal(wf, rf), (co, ci)
        if wd is not None:
            assert torch.equal(wd, rd), (co, ci)


def test_pack_weights_multi_bf16x3_equals_per_layer_pack():
    """BDN_BF16X3: the LDS-tiled packer of the regular layers (round 6) and the element-wise one of the rest write the same
    [w_hi | w_hi | w_lo] images as bdn_pack_weights(BDN_BF16X3) layer by layer."""
    import struct
    X3 = _lib.BDN_BF16X3
    layers = [(64, 13, 16, False), (64, 64, 64, True), (128, 64, 64, True), (256, 384, 384, True), (64, 128, 128, True), (32, 64, 64, False)]
    ws, outs, rec = [], [], b''
    for i, (co, ci, cip, has_wd) in enumerate(layers):
        w = dev(_rand((co, ci, 3, 3), 450 + i))
        wf = torch.full((co * 9 * 3 * cip,), 7.0, dtype=torch.bfloat16, device='cuda')
        wd = torch.full((co * 9 * 3 * cip,), 7.0, dtype=torch.bfloat16, device='cuda') if has_wd else None
        ws.append((w, co, ci, cip)); outs.append((wf, wd))
        rec += struct.pack('<QQQiiii', w.data_ptr(), wf.data_ptr(), wd.data_ptr() if has_wd else 0, co, ci, cip, 0)
    desc = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    _lib.call('bdn_pack_weights_multi', X3, desc.data_ptr(), len(ws), st())
    torch.cuda.synchronize()
    for (w, co, ci, cip), (wf, wd) in zip(ws, outs):
        rf = torch.empty_like(wf)
        rd = torch.empty_like(wd) if wd is not None else None
        _lib.call('bdn_pack_weights', X3, w.data_ptr(), rf.data_ptr(), rd.data_ptr() if rd is not None else None, co, ci, cip, st())
        torch.cuda.synchronize()
        assert torch.equal(wf, rf), (co, ci)
        if wd is not None:
            assert torch.equal(wd, rd), (co, ci)


@pytest.mark.parametrize('case', [(4, 32, 32, 2, 64), (6, 37, 50, 3, 64), (2, 128, 128, 1, 64), (4, 24, 16, 2, 80)])
def test_first_layer_wgrad_with_fused_bn_bwd(case):
    """bdn_bn_bwd_finalize + bdn_conv3x3_wgrad_bnbwd == bdn_bn_bwd_apply + bdn_conv3x3_wgrad (bf16, 13 real channels padded to
    16): same dgamma / dbeta / sums bit for bit, weight gradient equal up to the summation order of the partial tiles.
    Ragged maps, several statistic groups, and a dA that is a 64-channel slice of a wider tensor (ldA = 80)."""
    N, H, W, ipg, ldA = case
    Cout, C0, Creal = 64, 16, 13
    dt, td = DT['bf16']
    G = N // ipg
    lib = _lib.load()
    assert lib.bdn_conv3x3_wgrad_bnbwd_supported(dt, N, H, W, Cout, C0, ipg) == 1
    assert lib.bdn_conv3x3_wgrad_bnbwd_supported(dt, N, H, W, 128, C0, ipg) == 0
    dA_full = to_nhwc('bf16', rnd('bf16', _rand((N, ldA, H, W), 301)))
    z = rnd('bf16', _rand((N, Cout, H, W), 302))
    z_d = to_nhwc('bf16', z)
    x = rnd('bf16', _rand((N, C0, H, W), 303)); x[:, Creal:] = 0
    x_d = to_nhwc('bf16', x)
    bn = bn_table(G, Cout, 304)
    for g in range(G):                                           # statistics of z itself so the ReLU masks are mixed
        zg = z[g * ipg:(g + 1) * ipg].double()
        mean, var = zg.mean((0, 2, 3)), zg.var((0, 2, 3), unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        gamma = bn[g, 2].double() / bn[g, 1].double()
        bn[g, 0], bn[g, 1] = mean.float(), inv.float()
        bn[g, 2] = (gamma * inv).float()
        bn[g, 3] = (0.1 - mean * gamma * inv).float()
    bn_d = dev(bn)
    # BatchNorm-backward partial sums the way the producers leave them: rows of (sum g, sum g*z)
    y = torch.einsum('nchw,nc->nchw', z.float(), bn[:, 2].repeat_interleave(ipg, 0)) + bn[:, 3].repeat_interleave(ipg, 0)[:, :, None, None]
    gm = torch.where(y > 0, dA_full.float().cpu().permute(0, 3, 1, 2)[:, :Cout], torch.zeros(()))
    rows = 4
    part = torch.zeros(G * rows, 2, Cout)
    for g in range(G):
        for r in range(rows):
            sl = slice(g * ipg, (g + 1) * ipg)
            hs = slice(r * H // rows, (r + 1) * H // rows)
            part[g * rows + r, 0] = gm[sl, :, hs].double().sum((0, 2, 3)).float()
            part[g * rows + r, 1] = (gm[sl, :, hs].double() * z[sl, :, hs].double()).sum((0, 2, 3)).float()
    part_d = dev(part)
    wsz = lib.bdn_wgrad_workspace_bytes(N, H, W, Cout, C0, ipg) // 4
    out = {}
    for fused in (0, 1):
        sums = torch.full((G, 2, Cout), float('nan'), device='cuda')
        dg, db = torch.empty(Cout, device='cuda'), torch.empty(Cout, device='cuda')
        wpart = torch.empty(wsz, device='cuda')
        dw = torch.full((Cout, Creal, 3, 3), float('nan'), device='cuda')
        if fused:
            _lib.call('bdn_bn_bwd_finalize', bn_d.data_ptr(), G, Cout, part_d.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(),
                      db.data_ptr(), None, st())
            _lib.call('bdn_conv3x3_wgrad_bnbwd', dt, dA_full.data_ptr(), ldA, z_d.data_ptr(), bn_d.data_ptr(), sums.data_ptr(), ipg,
                      Cout, x_d.data_ptr(), C0, wpart.data_ptr(), dw.data_ptr(), Creal, N, H, W, st())
        else:
            dz = torch.empty(N, H, W, Cout, dtype=td, device='cuda')
            _lib.call('bdn_bn_bwd_apply', dt, dA_full.data_ptr(), ldA, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, Cout,
                      part_d.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dz.data_ptr(), None, st())
            _lib.call('bdn_conv3x3_wgrad', dt, dz.data_ptr(), Cout, x_d.data_ptr(), C0, None, 0, 0, None, ipg,
                      wpart.data_ptr(), dw.data_ptr(), Creal, N, H, W, st())
        torch.cuda.synchronize()
        out[fused] = (sums.cpu(), dg.cpu(), db.cpu(), dw.cpu())
    for i in range(3):
        assert torch.equal(out[0][i], out[1][i])
    assert torch.isfinite(out[1][3]).all()
    assert_close('fused first-layer dW', out[1][3], out[0][3], 2e-6)
    with pytest.raises(RuntimeError):
        _lib.call('bdn_conv3x3_wgrad_bnbwd', dt, dA_full.data_ptr(), ldA, z_d.data_ptr(), bn_d.data_ptr(), sums.data_ptr(), ipg,
                  128, x_d.data_ptr(), C0, wpart.data_ptr(), dw.data_ptr(), Creal, N, H, W, st())


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (1, 45, 22, 64), (2, 11, 11, 128), (1, 2, 3, 256)])
def test_product_pool_equals_separate_kernels(prec, shape):
    """bdn_product_pool == bdn_fuse_product + bdn_bnrelu_pool, bit for bit (odd sizes: floor-mode pooling)."""
    B, H, W, C = shape
    dt, td = DT[prec]
    z_d = to_nhwc(prec, rnd(prec, _rand((2 * B, C, H, W), 71)))
    bn_d = dev(bn_table(2, C, 72))
    f1 = torch.empty(B, H, W, C, dtype=td, device='cuda'); f2 = torch.full_like(f1, 3.0)
    p1 = torch.empty(2 * B, H // 2, W // 2, C, dtype=td, device='cuda'); p2 = torch.full_like(p1, 3.0)
    _lib.call('bdn_fuse_product', dt, z_d.data_ptr(), bn_d.data_ptr(), f1.data_ptr(), B, H, W, C, st())
    _lib.call('bdn_bnrelu_pool', dt, z_d.data_ptr(), bn_d.data_ptr(), B, p1.data_ptr(), 2 * B, H, W, C, st())
    _lib.call('bdn_product_pool', dt, z_d.data_ptr(), bn_d.data_ptr(), f2.data_ptr(), p2.data_ptr(), B, H, W, C, st())
    torch.cuda.synchronize()
    assert torch.equal(f1, f2) and torch.equal(p1, p2)


def test_wgrad_phases_variant_and_per_call_flags():
    """bdn_conv3x3_wgrad_ex(phases 1 then 2) == bdn_conv3x3_wgrad; the per-call grid-size field changes the workspace, not the
    result beyond the summation order of the partial tiles; bdn_conv3x3_wgrad_variant names the kernel family; there is no
    process-wide tuning state (the sizing query is a pure function of its arguments)."""
    from fabric_amd._lib import WG_SIMPLE, wg_flags
    lib = _lib.load()
    N, H, W, Cout, C0, ipg = 16, 32, 32, 128, 64, 8
    dt, td = DT['bf16']
    dz = to_nhwc('bf16', rnd('bf16', _rand((N, Cout, H, W), 91)))
    x = to_nhwc('bf16', rnd('bf16', _rand((N, C0, H, W), 92)))
    assert lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, C0, 0, ipg, IN_PLAIN, 0) == 5        # the role-split kernel
    assert lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, C0, 0, ipg, IN_BNRELU, 0) == 5       # also with BatchNorm on load
    assert lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, 16, 0, ipg, IN_PLAIN, 0) == WG_SIMPLE  # narrow input
    assert lib.bdn_conv3x3_wgrad_variant(dt, N, 8, 8, Cout, C0, 0, ipg, IN_PLAIN, 0) == WG_SIMPLE  # 8x8 maps
    assert lib.bdn_conv3x3_wgrad_variant(DT['fp32'][0], N, H, W, Cout, C0, 0, ipg, IN_PLAIN, 0) == WG_SIMPLE
    assert lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, C0, 0, ipg, IN_PLAIN, wg_flags(kernel=WG_SIMPLE)) == WG_SIMPLE
    default_bytes = lib.bdn_wgrad_workspace_bytes(N, H, W, Cout, C0, ipg)
    res = {}
    for blocks in (0, 64):
        fl = wg_flags(0, 0, blocks)
        nbytes = lib.bdn_wgrad_workspace_bytes_ex(dt, N, H, W, Cout, C0, 0, ipg, IN_PLAIN, fl)
        part = torch.empty(nbytes // 4, device='cuda')
        a = torch.empty(Cout, C0, 3, 3, device='cuda')
        b = torch.full_like(a, float('nan'))
        _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, x.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg,
                  part.data_ptr(), a.data_ptr(), C0, N, H, W, fl | 3, st())
        for ph in (1, 2):
            _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, x.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg,
                      part.data_ptr(), b.data_ptr(), C0, N, H, W, fl | ph, st())
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        res[blocks] = (a.cpu(), nbytes)
    assert res[64][1] < res[0][1] <= default_bytes
    assert lib.bdn_wgrad_workspace_bytes(N, H, W, Cout, C0, ipg) == default_bytes        # nothing above changed what it answers
    c = torch.empty(Cout, C0, 3, 3, device='cuda')
    part = torch.empty(default_bytes // 4, device='cuda')
    _lib.call('bdn_conv3x3_wgrad', dt, dz.data_ptr(), Cout, x.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg,
              part.data_ptr(), c.data_ptr(), C0, N, H, W, st())
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), res[0][0])
    assert_close('dw across grid sizes', res[64][0], res[0][0], 1e-5)
    with pytest.raises(RuntimeError):
        _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, x.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg,
                  part.data_ptr(), c.data_ptr(), C0, N, H, W, 0, st())


@pytest.mark.parametrize('mode', ['bnrelu', 'plain', 'plain2'])
def test_wgrad_kernel_variants_agree(mode):
    """Both weight-gradient kernels a shape may run give the same dW up to the summation order of the partial tiles
    (ragged map, two statistic groups / two concatenated sources).  The role-split kernel stages the patch through its
    producer waves' registers with BatchNorm+ReLU ('bnrelu') or by LDS-DMA ('plain'): a plain operand that already holds
    relu(bn(z)) (bdn_bnrelu) must give the BatchNorm-on-load result bit for bit."""
    from fabric_amd._lib import WG_ROLE, WG_SIMPLE, wg_flags
    lib = _lib.load()
    N, H, W, Cout, ipg = 6, 37, 50, 128, 3
    dt, td = DT['bf16']
    dz = to_nhwc('bf16', rnd('bf16', _rand((N, Cout, H, W), 195)))
    if mode == 'bnrelu':
        C0, C1 = 128, 0
        x0, x1 = to_nhwc('bf16', rnd('bf16', _rand((N, C0, H, W), 196))), None
        bn_d, in_mode = dev(bn_table(N // ipg, C0, 197)), IN_BNRELU
    else:
        C0, C1 = (64, 64) if mode == 'plain2' else (128, 0)
        x0 = to_nhwc('bf16', rnd('bf16', _rand((N, C0, H, W), 196)))
        x1 = to_nhwc('bf16', rnd('bf16', _rand((N, C1, H, W), 198))) if C1 else None
        bn_d, in_mode = None, IN_PLAIN
    out, ran = {}, {}
    for v in (WG_SIMPLE, WG_ROLE):
        fl = wg_flags(3, v, 0)
        ran[v] = lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, C0, C1, ipg, in_mode, fl)
        part = torch.empty(lib.bdn_wgrad_workspace_bytes_ex(dt, N, H, W, Cout, C0, C1, ipg, in_mode, fl) // 4, device='cuda')
        dw = torch.full((Cout, C0 + C1, 3, 3), float('nan'), device='cuda')
        _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, x0.data_ptr(), C0, x1.data_ptr() if x1 is not None else None, C1,
                  in_mode, bn_d.data_ptr() if bn_d is not None else None, ipg, part.data_ptr(), dw.data_ptr(), C0 + C1, N, H, W, fl, st())
        torch.cuda.synchronize()
        out[v] = dw.cpu()
        assert torch.isfinite(out[v]).all()
    assert ran[WG_SIMPLE] == WG_SIMPLE and ran[WG_ROLE] == WG_ROLE
    assert_close('role-split vs simple', out[WG_ROLE], out[WG_SIMPLE], 2e-6)
    if mode == 'bnrelu':                                     # register path (BatchNorm on load) == LDS-DMA path on the materialised operand
        act = torch.empty_like(x0)
        _lib.call('bdn_bnrelu', dt, x0.data_ptr(), bn_d.data_ptr(), ipg, act.data_ptr(), N, H, W, C0, st())
        fl = wg_flags(3, WG_ROLE, 0)
        part = torch.empty(lib.bdn_wgrad_workspace_bytes_ex(dt, N, H, W, Cout, C0, 0, ipg, IN_PLAIN, fl) // 4, device='cuda')
        dw = torch.full((Cout, C0, 3, 3), float('nan'), device='cuda')
        _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, act.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg,
                  part.data_ptr(), dw.data_ptr(), C0, N, H, W, fl, st())
        torch.cuda.synchronize()
        assert torch.equal(dw.cpu(), out[WG_ROLE])

@pytest.mark.parametrize('shape', [(1, 8, 16, 64, 64, 1), (1, 9, 17, 64, 64, 1), (2, 8, 16, 128, 64, 1), (3, 24, 16, 64, 192, 3), (4, 16, 32, 64, 64, 2)])
@pytest.mark.parametrize('bn', [False, True])
def test_role_split_wgrad_on_tiny_problems(shape, bn):
    """wgrad7's pipeline prologue / epilogue on problems of one to a few 128-pixel chunks per block (a block whose split holds a single
    chunk still issues its two look-ahead fetches, all masked) and on odd chunk counts: same dW as the one-chunk-at-a-time kernel."""
    from fabric_amd._lib import WG_ROLE, WG_SIMPLE, wg_flags
    lib = _lib.load()
    N, H, W, Cout, C0, ipg = shape
    dt, td = DT['bf16']
    dz = to_nhwc('bf16', rnd('bf16', _rand((N, Cout, H, W), 301)))
    x0 = to_nhwc('bf16', rnd('bf16', _rand((N, C0, H, W), 302)))
    bn_d = dev(bn_table(N // ipg, C0, 303)) if bn else None
    mode = IN_BNRELU if bn else IN_PLAIN
    out = {}
    for v in (WG_SIMPLE, WG_ROLE):
        for blocks in ((0, 1, 7) if v == WG_ROLE else (0,)):          # 1 block: one split holds every chunk; 7: odd chunk counts per split
            fl = wg_flags(3, v, blocks)
            assert lib.bdn_conv3x3_wgrad_variant(dt, N, H, W, Cout, C0, 0, ipg, mode, fl) == v
            part = torch.empty(lib.bdn_wgrad_workspace_bytes_ex(dt, N, H, W, Cout, C0, 0, ipg, mode, fl) // 4, device='cuda')
            dw = torch.full((Cout, C0, 3, 3), float('nan'), device='cuda')
            _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), Cout, x0.data_ptr(), C0, None, 0, mode, bn_d.data_ptr() if bn else None, ipg,
                      part.data_ptr(), dw.data_ptr(), C0, N, H, W, fl, st())
            torch.cuda.synchronize()
            out[(v, blocks)] = dw.cpu()
            assert torch.isfinite(out[(v, blocks)]).all()
    for blocks in (0, 1, 7):
        assert_close(f'role-split ({blocks} blocks) vs simple', out[(WG_ROLE, blocks)], out[(WG_SIMPLE, 0)], 2e-6)


# ------------------------------------------------------------------ bf16x3: float32 tensors, bf16 hi/lo split GEMM operands
def _split_ref(t_nchw):
    """[N,C,H,W] f32 -> the [N,H,W,2C] bf16 operand bdn_split_pack must produce (hi | lo)."""
    hi = t_nchw.to(torch.bfloat16)
    lo = (t_nchw - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], 1).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('case', [(2, 9, 7, 16, 0, False), (4, 16, 16, 64, 64, False), (4, 12, 20, 128, 0, True)])
def test_split_pack_is_the_exact_hi_lo_split(case):
    """bdn_split_pack: hi = bf16(x), lo = bf16(x - hi) of cat(src0, src1) resp. relu(bn(src0)); hi + lo reproduces x to 2^-16."""
    N, H, W, C0, C1, use_bn = case
    ipg = N // 2
    x0, x1 = _rand((N, C0, H, W), 301), (_rand((N, C1, H, W), 302) if C1 else None)
    bn = bn_table(2, C0, 303)
    a0 = bnrelu_ref('fp32', x0, bn, ipg) if use_bn else x0
    full = torch.cat([a0, x1], 1) if C1 else a0
    out = torch.empty(N, H, W, 2 * (C0 + C1), dtype=torch.bfloat16, device='cuda')
    d0, d1 = to_nhwc('fp32', x0), (to_nhwc('fp32', x1) if C1 else None)
    bn_d = dev(bn)
    _lib.call('bdn_split_pack', d0.data_ptr(), C0, d1.data_ptr() if C1 else None, C1, IN_BNRELU if use_bn else IN_PLAIN,
              bn_d.data_ptr(), ipg, out.data_ptr(), N, H, W, st())
    torch.cuda.synchronize()
    got = out.cpu()
    if use_bn:      # the kernel's fma vs the reference's mul + add: compare the reconstructed value
        rec = (got[..., :C0 + C1].float() + got[..., C0 + C1:].float()).permute(0, 3, 1, 2)
        assert_close('hi+lo', rec, full, 2e-5)
    else:
        assert torch.equal(got, _split_ref(full))
        rec = (got[..., :C0 + C1].float() + got[..., C0 + C1:].float()).permute(0, 3, 1, 2)
        assert (rec - full).abs().max() <= 2.0 ** -16 * full.abs().max()


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, 1), (2, 8, 8, 128, 64, 2), (1, 22, 45, 16, 64, 1), (3, 19, 33, 64, 128, 3),
                                  (2, 12, 12, 192, 64, 2), (36, 32, 32, 128, 256, 18), (2, 8, 8, 256, 128, 1)])
def test_conv3x3_bf16x3_forward_dgrad_wgrad(case):
    """The bf16 kernels on split operands (three times the reduction length) against the float32 oracle: forward with
    statistics, data gradient and weight gradient all within 1e-4 of the tensor's magnitude (a plain bf16 GEMM sits at 1e-2)."""
    N, H, W, Cin, Cout, ipg = case
    lib = _lib.load()
    X3 = _lib.BDN_BF16X3
    x = _rand((N, Cin, H, W), 311)
    w = _rand((Cout, Cin, 3, 3), 312) * 0.1
    b = _rand((Cout,), 313)
    ref = F.conv2d(x, w, b, padding=1)
    # forward
    sp = torch.empty(N, H, W, 2 * Cin, dtype=torch.bfloat16, device='cuda')
    xd = to_nhwc('fp32', x)
    _lib.call('bdn_split_pack', xd.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg, sp.data_ptr(), N, H, W, st())
    wf = torch.empty(Cout, 9, 3 * Cin, dtype=torch.bfloat16, device='cuda')
    wd = torch.empty(Cin, 9, 3 * Cout, dtype=torch.bfloat16, device='cuda') if Cin % 64 == 0 else None
    wdev = dev(w)
    _lib.call('bdn_pack_weights', X3, wdev.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else None, Cout, Cin, Cin, st())
    out = torch.full((N, H, W, Cout), float('nan'), device='cuda')
    nt = lib.bdn_conv3x3_num_mtiles_ex(X3, N, H, W, Cin, Cout, ipg)         # (operands of >= 64 channels: the fused-split-product kernels' own tile plan)
    stats = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    bd = dev(b)
    _lib.call('bdn_conv3x3', X3, sp.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg, wf.data_ptr(), bd.data_ptr(), out.data_ptr(),
              stats.data_ptr(), N, H, W, Cout, st())
    torch.cuda.synchronize()
    assert_close('x3 fwd', from_nhwc(out), ref, 1e-4)
    G = N // ipg
    ssum = stats[:, 0].cpu().double().reshape(G, -1, Cout).sum(1)
    want = ref.double().reshape(G, ipg, Cout, H * W).sum((1, 3))
    assert (ssum - want).abs().max() <= 1e-4 * want.abs().max() + 1e-3
    with pytest.raises(RuntimeError):          # the split operand carries cat and BatchNorm already
        _lib.call('bdn_conv3x3', X3, sp.data_ptr(), Cin, None, 0, IN_BNRELU, bd.data_ptr(), ipg, wf.data_ptr(), bd.data_ptr(),
                  out.data_ptr(), None, N, H, W, Cout, st())
    # gradients
    dz = _rand((N, Cout, H, W), 314)
    dzd = to_nhwc('fp32', dz)
    spd = torch.empty(N, H, W, 2 * Cout, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', dzd.data_ptr(), Cout, None, 0, IN_PLAIN, None, ipg, spd.data_ptr(), N, H, W, st())
    if wd is not None:
        dA = torch.full((N, H, W, Cin), float('nan'), device='cuda')
        _lib.call('bdn_conv3x3', X3, spd.data_ptr(), Cout, None, 0, IN_PLAIN, None, ipg, wd.data_ptr(), None, dA.data_ptr(), None,
                  N, H, W, Cin, st())
        torch.cuda.synchronize()
        assert_close('x3 dgrad', from_nhwc(dA), torch.nn.grad.conv2d_input(x.shape, w, dz, padding=1), 1e-4)
    nb = lib.bdn_wgrad_workspace_bytes_ex(X3, N, H, W, Cout, Cin, 0, ipg, IN_PLAIN, 3)
    assert nb <= lib.bdn_wgrad_workspace_bytes(N, H, W, Cout, Cin, ipg)
    part = torch.empty(nb // 4, device='cuda')
    dw = torch.full((Cout, Cin, 3, 3), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3_wgrad', X3, spd.data_ptr(), Cout, sp.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg,
              part.data_ptr(), dw.data_ptr(), Cin, N, H, W, st())
    torch.cuda.synchronize()
    assert_close('x3 wgrad', dw.cpu(), torch.nn.grad.conv2d_weight(x, w.shape, dz, padding=1), 1e-4)


@pytest.mark.parametrize('dtype', ['x3', 'x2'])
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, 1, True), (4, 8, 8, 128, 64, 2, True), (3, 19, 33, 64, 128, 3, True), (2, 12, 12, 192, 64, 2, False),
                                  (36, 32, 32, 128, 256, 18, True), (2, 8, 8, 512, 128, 1, True), (2, 40, 72, 64, 64, 1, True)])
def test_conv3x3_x3src_equals_split_pack_then_conv(case, dtype):
    """bdn_conv3x3_x3src (float32 operand; BatchNorm+ReLU and the bf16 hi / lo split inside the convolution's staging -- models/unet_parts.py:14-16)
    against the two launches it replaces, bdn_split_pack + bdn_conv3x3 on the split operand: output, statistics partials and the split operand it
    leaves for the weight-gradient GEMM are all bit-identical; and the float32 oracle holds within 1e-4 (three terms)."""
    N, H, W, Cin, Cout, ipg, use_bn = case
    lib = _lib.load()
    DT = _lib.BDN_BF16X3 if dtype == 'x3' else _lib.BDN_BF16X2
    x = _rand((N, Cin, H, W), 711)
    w = _rand((Cout, Cin, 3, 3), 712) * 0.1
    b = _rand((Cout,), 713)
    bn = bn_table(N // ipg, Cin, 714)
    xd, bn_d, bd, wdev = to_nhwc('fp32', x), dev(bn), dev(b), dev(w)
    mode = IN_BNRELU if use_bn else IN_PLAIN
    wf = torch.empty(Cout, 9, 3 * Cin, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_pack_weights', _lib.BDN_BF16X3, wdev.data_ptr(), wf.data_ptr(), None, Cout, Cin, Cin, st())
    nt = lib.bdn_conv3x3_num_mtiles_ex(DT, N, H, W, Cin, Cout, ipg)
    # the two launches
    sp = torch.empty(N, H, W, 2 * Cin, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', xd.data_ptr(), Cin, None, 0, mode, bn_d.data_ptr(), ipg, sp.data_ptr(), N, H, W, st())
    out0 = torch.full((N, H, W, Cout), float('nan'), device='cuda')
    st0 = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3', DT, sp.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg, wf.data_ptr(), bd.data_ptr(), out0.data_ptr(),
              st0.data_ptr(), N, H, W, Cout, st())
    # the one launch
    sp1 = torch.full((N, H, W, 2 * Cin), float('nan'), dtype=torch.bfloat16, device='cuda')
    out1 = torch.full((N, H, W, Cout), float('nan'), device='cuda')
    st1 = torch.full((nt, 2, Cout), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3_x3src', DT, xd.data_ptr(), Cin, mode, bn_d.data_ptr(), ipg, wf.data_ptr(), bd.data_ptr(), out1.data_ptr(),
              st1.data_ptr(), sp1.data_ptr(), N, H, W, Cout, st())
    torch.cuda.synchronize()
    assert torch.equal(sp1.view(torch.int16), sp.view(torch.int16))
    assert torch.equal(out1, out0) and torch.equal(st1, st0)
    # without the by-product (eval forwards), same output
    out2 = torch.full((N, H, W, Cout), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3_x3src', DT, xd.data_ptr(), Cin, mode, bn_d.data_ptr(), ipg, wf.data_ptr(), bd.data_ptr(), out2.data_ptr(),
              None, None, N, H, W, Cout, st())
    torch.cuda.synchronize()
    assert torch.equal(out2, out0)
    if dtype == 'x3':
        a = bnrelu_ref('fp32', x, bn, ipg) if use_bn else x
        assert_close('x3src fwd', from_nhwc(out1), F.conv2d(a, w, b, padding=1), 1e-4)
    with pytest.raises(RuntimeError):          # operands the staging table does not hold
        _lib.call('bdn_conv3x3_x3src', DT, xd.data_ptr(), 1024, mode, bn_d.data_ptr(), ipg, wf.data_ptr(), bd.data_ptr(), out2.data_ptr(),
                  None, None, N, H, W, Cout, st())


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, 1), (2, 8, 8, 128, 64, 2), (3, 19, 33, 64, 128, 3), (2, 12, 12, 192, 64, 2)])
def test_conv3x3_bf16x2_backward_gemms(case):
    """BDN_BF16X2, the two-term backward of the bf16x3 setting, on the operands and filter images of BDN_BF16X3: the data gradient equals
    the exact data gradient with the FILTER rounded to bf16 (dz in full), the weight gradient the exact one with DZ rounded to bf16 (the
    activations in full) -- each within 1e-4 of the tensor's magnitude (autograd of models/unet_parts.py:13,16)."""
    N, H, W, Cin, Cout, ipg = case
    lib = _lib.load()
    X3, X2 = _lib.BDN_BF16X3, _lib.BDN_BF16X2
    x = _rand((N, Cin, H, W), 321)
    w = _rand((Cout, Cin, 3, 3), 322) * 0.1
    dz = _rand((N, Cout, H, W), 323)
    sp = torch.empty(N, H, W, 2 * Cin, dtype=torch.bfloat16, device='cuda')
    xd = to_nhwc('fp32', x)
    _lib.call('bdn_split_pack', xd.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg, sp.data_ptr(), N, H, W, st())
    wf = torch.empty(Cout, 9, 3 * Cin, dtype=torch.bfloat16, device='cuda')
    wd = torch.empty(Cin, 9, 3 * Cout, dtype=torch.bfloat16, device='cuda')
    wdev = dev(w)
    _lib.call('bdn_pack_weights', X3, wdev.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin, Cin, st())
    dzd = to_nhwc('fp32', dz)
    spd = torch.empty(N, H, W, 2 * Cout, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', dzd.data_ptr(), Cout, None, 0, IN_PLAIN, None, ipg, spd.data_ptr(), N, H, W, st())
    dA = torch.full((N, H, W, Cin), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3', X2, spd.data_ptr(), Cout, None, 0, IN_PLAIN, None, ipg, wd.data_ptr(), None, dA.data_ptr(), None,
              N, H, W, Cin, st())
    torch.cuda.synchronize()
    w_hi, dz_hi = w.to(torch.bfloat16).float(), dz.to(torch.bfloat16).float()
    assert_close('x2 dgrad', from_nhwc(dA), torch.nn.grad.conv2d_input(x.shape, w_hi, dz, padding=1), 1e-4)
    # ... and it is NOT the three-term product: the full-precision filter is further away than the rounded one
    full = torch.nn.grad.conv2d_input(x.shape, w, dz, padding=1)
    assert (from_nhwc(dA) - full).abs().max() > 3e-4 * full.abs().max()
    nb = lib.bdn_wgrad_workspace_bytes_ex(X2, N, H, W, Cout, Cin, 0, ipg, IN_PLAIN, 3)
    part = torch.empty(nb // 4, device='cuda')
    dw = torch.full((Cout, Cin, 3, 3), float('nan'), device='cuda')
    _lib.call('bdn_conv3x3_wgrad', X2, spd.data_ptr(), Cout, sp.data_ptr(), Cin, None, 0, IN_PLAIN, None, ipg,
              part.data_ptr(), dw.data_ptr(), Cin, N, H, W, st())
    torch.cuda.synchronize()
    assert_close('x2 wgrad', dw.cpu(), torch.nn.grad.conv2d_weight(x, w.shape, dz_hi, padding=1), 1e-4)


@pytest.mark.parametrize('dtype', ['x3', 'x2'])
@pytest.mark.parametrize('case', [(4, 32, 32, 2, 64), (6, 37, 50, 3, 64), (2, 128, 128, 1, 64), (4, 24, 16, 2, 80)])
def test_first_layer_wgrad_with_fused_bn_bwd_bf16x3(case, dtype):
    """bf16x3 / bf16x2 form of bdn_conv3x3_wgrad_bnbwd (round 6): float32 dA and z, BatchNorm backward + hi / lo split of dz inside the staging,
    the input's split operand from bdn_pack_input(BDN_BF16X3)'s layout, the terms of the split product summed in one accumulator.  Against
    bdn_bn_bwd_apply_split + bdn_conv3x3_wgrad(BDN_BF16X3 / X2) on the same inputs (dgamma / dbeta / sums bit for bit, the weight gradient up
    to summation order) and, three terms, against the exact float32 weight gradient within 1e-4."""
    from fabric_amd._lib import BDN_F32, BDN_BF16X3, BDN_BF16X2
    N, H, W, ipg, ldA = case
    Cout, C0, Creal = 64, 16, 13
    DT_ = BDN_BF16X3 if dtype == 'x3' else BDN_BF16X2
    G = N // ipg
    lib = _lib.load()
    assert lib.bdn_conv3x3_wgrad_bnbwd_supported(DT_, N, H, W, Cout, C0, ipg) == 1
    dA_full = to_nhwc('fp32', _rand((N, ldA, H, W), 801))
    z = _rand((N, Cout, H, W), 802)
    z_d = to_nhwc('fp32', z)
    x = _rand((N, C0, H, W), 803); x[:, Creal:] = 0
    x_d = to_nhwc('fp32', x)
    xs = torch.empty(N, H, W, 2 * C0, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', x_d.data_ptr(), C0, None, 0, IN_PLAIN, None, ipg, xs.data_ptr(), N, H, W, st())
    bn = bn_table(G, Cout, 804)
    for g in range(G):
        zg = z[g * ipg:(g + 1) * ipg].double()
        mean, var = zg.mean((0, 2, 3)), zg.var((0, 2, 3), unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        gamma = bn[g, 2].double() / bn[g, 1].double()
        bn[g, 0], bn[g, 1] = mean.float(), inv.float()
        bn[g, 2] = (gamma * inv).float()
        bn[g, 3] = (0.1 - mean * gamma * inv).float()
    bn_d = dev(bn)
    y = torch.einsum('nchw,nc->nchw', z, bn[:, 2].repeat_interleave(ipg, 0)) + bn[:, 3].repeat_interleave(ipg, 0)[:, :, None, None]
    gm = torch.where(y > 0, dA_full.cpu().permute(0, 3, 1, 2)[:, :Cout], torch.zeros(()))
    rows = 4
    part = torch.zeros(G * rows, 2, Cout)
    for g in range(G):
        for r in range(rows):
            sl = slice(g * ipg, (g + 1) * ipg)
            hs = slice(r * H // rows, (r + 1) * H // rows)
            part[g * rows + r, 0] = gm[sl, :, hs].double().sum((0, 2, 3)).float()
            part[g * rows + r, 1] = (gm[sl, :, hs].double() * z[sl, :, hs].double()).sum((0, 2, 3)).float()
    part_d = dev(part)
    wsz = max(lib.bdn_wgrad_workspace_bytes(N, H, W, Cout, C0, ipg), lib.bdn_wgrad_workspace_bytes_ex(DT_, N, H, W, Cout, C0, 0, ipg, IN_PLAIN, 3)) // 4
    out = {}
    for fused in (0, 1):
        sums = torch.full((G, 2, Cout), float('nan'), device='cuda')
        dg, db = torch.empty(Cout, device='cuda'), torch.empty(Cout, device='cuda')
        wpart = torch.empty(wsz, device='cuda')
        dw = torch.full((Cout, Creal, 3, 3), float('nan'), device='cuda')
        if fused:
            _lib.call('bdn_bn_bwd_finalize', bn_d.data_ptr(), G, Cout, part_d.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(),
                      db.data_ptr(), None, st())
            _lib.call('bdn_conv3x3_wgrad_bnbwd', DT_, dA_full.data_ptr(), ldA, z_d.data_ptr(), bn_d.data_ptr(), sums.data_ptr(), ipg,
                      Cout, xs.data_ptr(), C0, wpart.data_ptr(), dw.data_ptr(), Creal, N, H, W, st())
        else:
            dzs = torch.empty(N, H, W, 2 * Cout, dtype=torch.bfloat16, device='cuda')
            _lib.call('bdn_bn_bwd_apply_split', dA_full.data_ptr(), ldA, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, Cout,
                      part_d.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dzs.data_ptr(), None, st())
            _lib.call('bdn_conv3x3_wgrad', DT_, dzs.data_ptr(), Cout, xs.data_ptr(), C0, None, 0, 0, None, ipg,
                      wpart.data_ptr(), dw.data_ptr(), Creal, N, H, W, st())
        torch.cuda.synchronize()
        out[fused] = (sums.cpu(), dg.cpu(), db.cpu(), dw.cpu())
    for i in range(3):
        assert torch.equal(out[0][i], out[1][i])
    assert torch.isfinite(out[1][3]).all()
    scale = out[0][3].abs().max().item()
    assert (out[1][3] - out[0][3]).abs().max().item() <= 2e-5 * scale          # same products, another summation order
    if dtype == 'x3':                                      # exact float32 reference: dz by bn_bwd_apply's formula, then autograd's weight gradient
        M = ipg * H * W
        dz = torch.empty_like(z)
        for g in range(G):
            sl = slice(g * ipg, (g + 1) * ipg)
            s0 = gm[sl].double().sum((0, 2, 3)) / M
            s1 = (gm[sl].double() * ((z[sl].double() - bn[g, 0].double()[None, :, None, None]) * bn[g, 1].double()[None, :, None, None])).sum((0, 2, 3)) / M
            xhat = (z[sl].double() - bn[g, 0].double()[None, :, None, None]) * bn[g, 1].double()[None, :, None, None]
            dz[sl] = (bn[g, 2].double()[None, :, None, None] * (gm[sl].double() - s0[None, :, None, None] - xhat * s1[None, :, None, None])).float()
        ref = torch.nn.grad.conv2d_weight(x[:, :Creal], (Cout, Creal, 3, 3), dz, padding=1)
        assert_close('x3 fused first-layer wgrad', out[1][3], ref, 2e-4)


def test_pack_input_and_head_bn_bwd_store_the_split_operand_directly():
    """bf16x3 setting, round 6: bdn_pack_input(BDN_BF16X3) and bdn_outc_bn_bwd_apply(BDN_BF16X3) store what bdn_split_pack would make of their
    float32 outputs -- the first convolution's operand and d4b's dz -- bit for bit, so a bf16x3 step launches no split pass for them."""
    from fabric_amd._lib import BDN_F32, BDN_BF16X3
    lib = _lib.load()
    B, C, H, W, Cp = 3, 13, 20, 24, 16
    a, b = dev(_rand((B, C, H, W), 91)), dev(_rand((B, C, H, W), 92))
    x0 = torch.empty(2 * B, H, W, Cp, device='cuda')
    _lib.call('bdn_pack_input', BDN_F32, a.data_ptr(), b.data_ptr(), x0.data_ptr(), B, C, H, W, Cp, st())
    ref = torch.empty(2 * B, H, W, 2 * Cp, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', x0.data_ptr(), Cp, None, 0, IN_PLAIN, None, B, ref.data_ptr(), 2 * B, H, W, st())
    got = torch.full_like(ref, float('nan'))
    _lib.call('bdn_pack_input', BDN_BF16X3, a.data_ptr(), b.data_ptr(), got.data_ptr(), B, C, H, W, Cp, st())
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # head: dz of the layer in front of the classifier
    B, H, W, C, ncls = 2, 33, 20, 64, 2
    z_d = to_nhwc('fp32', _rand((B, C, H, W), 93))
    bn_d = dev(bn_table(1, C, 94))
    w_d, dl_d = dev(_rand((ncls, C), 95, 0.2)), dev(_rand((B, ncls, H, W), 96))
    rows = lib.bdn_outc_bwd_rows(BDN_F32, B, H, W, C)
    dw, db = torch.empty(ncls, C, device='cuda'), torch.empty(ncls, device='cuda')
    part = torch.empty(rows, 2, C, device='cuda')
    ows = torch.empty(lib.bdn_outc_bwd_workspace_bytes(BDN_F32, B, H, W, C, ncls) // 4, device='cuda')
    _lib.call('bdn_outc_bwd', BDN_F32, dl_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), w_d.data_ptr(), None, dw.data_ptr(), db.data_ptr(),
              part.data_ptr(), ows.data_ptr(), B, H, W, C, ncls, st())
    sums = torch.empty(1, 2, C, device='cuda'); dg, dbt = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    _lib.call('bdn_bn_bwd_finalize', bn_d.data_ptr(), 1, C, part.data_ptr(), rows, 1, sums.data_ptr(), dg.data_ptr(), dbt.data_ptr(), None, st())
    dz = torch.empty(B, H, W, C, device='cuda')
    _lib.call('bdn_outc_bn_bwd_apply', BDN_F32, dl_d.data_ptr(), w_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), B, sums.data_ptr(),
              dz.data_ptr(), B, H, W, C, ncls, st())
    ref = torch.empty(B, H, W, 2 * C, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', dz.data_ptr(), C, None, 0, IN_PLAIN, None, B, ref.data_ptr(), B, H, W, st())
    got = torch.full_like(ref, float('nan'))
    _lib.call('bdn_outc_bn_bwd_apply', BDN_BF16X3, dl_d.data_ptr(), w_d.data_ptr(), z_d.data_ptr(), bn_d.data_ptr(), B, sums.data_ptr(),
              got.data_ptr(), B, H, W, C, ncls, st())
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_split_outputs_of_the_bf16x3_producers_equal_split_pack():
    """bf16x3 setting: bdn_product_pool_split / bdn_upsample2x_split / bdn_bn_bwd_apply_split store their float32 results directly as
    the [hi | lo] bf16 operands of the consuming GEMMs; each must equal bdn_split_pack of the float32 kernel's output bit for bit."""
    from fabric_amd._lib import BDN_F32
    B, H, W, C, Cu = 2, 24, 20, 64, 128
    z = _rand((2 * B, C, H, W), 81)
    bn = bn_table(2, C, 82)
    z_d, bn_d = to_nhwc('fp32', z), dev(bn)
    f = torch.empty(B, H, W, C, device='cuda'); pool = torch.empty(2 * B, H // 2, W // 2, C, device='cuda')
    _lib.call('bdn_product_pool', BDN_F32, z_d.data_ptr(), bn_d.data_ptr(), f.data_ptr(), pool.data_ptr(), B, H, W, C, st())
    # upsampled map of a (B, H/2, W/2, Cu) source with BatchNorm+ReLU on load: second source of the decoder operand [f | U]
    src = to_nhwc('fp32', _rand((B, Cu, H // 2, W // 2), 83)); bnu = dev(bn_table(1, Cu, 84))
    U = torch.empty(B, H, W, Cu, device='cuda')
    _lib.call('bdn_upsample2x', BDN_F32, src.data_ptr(), IN_BNRELU, bnu.data_ptr(), U.data_ptr(), B, H // 2, W // 2, H, W, Cu, st())
    Ct = C + Cu
    ref_cat = torch.empty(B, H, W, 2 * Ct, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', f.data_ptr(), C, U.data_ptr(), Cu, IN_PLAIN, None, B, ref_cat.data_ptr(), B, H, W, st())
    ref_pool = torch.empty(2 * B, H // 2, W // 2, 2 * C, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', pool.data_ptr(), C, None, 0, IN_PLAIN, None, B, ref_pool.data_ptr(), 2 * B, H // 2, W // 2, st())
    got_cat = torch.zeros_like(ref_cat); got_pool = torch.zeros_like(ref_pool)
    _lib.call('bdn_product_pool_split', z_d.data_ptr(), bn_d.data_ptr(), got_cat.data_ptr(), 2 * Ct, Ct, got_pool.data_ptr(), B, H, W, C, st())
    _lib.call('bdn_upsample2x_split', src.data_ptr(), IN_BNRELU, bnu.data_ptr(), got_cat.data_ptr(), 2 * Ct, C, Ct, B, H // 2, W // 2, H, W, Cu, st())
    torch.cuda.synchronize()
    assert torch.equal(got_pool, ref_pool)
    assert torch.equal(got_cat, ref_cat)
    # BatchNorm backward with the split dz
    lib = _lib.load()
    N, ipg = 2 * B, B
    dA = to_nhwc('fp32', _rand((N, C, H, W), 85))
    ws = torch.empty(lib.bdn_bn_bwd_workspace_bytes(BDN_F32, N, H, W, C, ipg) // 4, device='cuda')
    sums = torch.empty(2, 2, C, device='cuda'); dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    dz = torch.empty(N, H, W, C, device='cuda')
    _lib.call('bdn_bn_bwd', BDN_F32, dA.data_ptr(), C, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, C, ws.data_ptr(), sums.data_ptr(), dg.data_ptr(),
              db.data_ptr(), dz.data_ptr(), st())
    ref_dz = torch.empty(N, H, W, 2 * C, dtype=torch.bfloat16, device='cuda')
    _lib.call('bdn_split_pack', dz.data_ptr(), C, None, 0, IN_PLAIN, None, ipg, ref_dz.data_ptr(), N, H, W, st())
    # the same partial rows the fused producers would leave: one row per group holding the finalized sums is not available here, so feed
    # the reduction's own block partials (raw_moment = 0: second moment already against xhat)
    G = N // ipg
    rows = (ws.numel() * 4 - lib.bdn_bn_bwd_scratch_bytes(G, C)) // (2 * C * 4) // G
    sums2 = torch.empty_like(sums); got_dz = torch.zeros_like(ref_dz)
    scratch = torch.empty(max(lib.bdn_bn_bwd_scratch_bytes(G, C) // 8, 1), dtype=torch.float64, device='cuda')
    _lib.call('bdn_bn_bwd_apply_split', dA.data_ptr(), C, z_d.data_ptr(), bn_d.data_ptr(), ipg, N, H, W, C, ws.data_ptr(), rows, 0,
              sums2.data_ptr(), dg.data_ptr(), db.data_ptr(), got_dz.data_ptr(), scratch.data_ptr(), st())
    torch.cuda.synchronize()
    assert torch.equal(sums2, sums)
    assert torch.equal(got_dz, ref_dz)
