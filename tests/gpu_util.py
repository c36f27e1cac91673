"""Helpers shared by the -m gpu parity tests: layout conversion between the oracle's NCHW float32
CPU tensors and the library's NHWC device tensors, and error summaries."""
import numpy as np
import torch

from fabric_amd import _lib
from fabric_amd._lib import BDN_BF16, BDN_F32

DT = {'fp32': (BDN_F32, torch.float32), 'bf16': (BDN_BF16, torch.bfloat16)}


def rnd(precision, t):
    """Round a float32 CPU tensor to the precision's storage type (and back to float32)."""
    return t.to(torch.bfloat16).float() if precision == 'bf16' else t.float()


def to_nhwc(precision, t_nchw):
    """NCHW float32 CPU -> NHWC device tensor of the precision's storage type."""
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(DT[precision][1]).cuda()


def from_nhwc(t):
    """NHWC device tensor -> NCHW float32 CPU."""
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def dev(t, dtype=torch.float32):
    return t.to(dtype).contiguous().cuda()


def st():
    return _lib.stream_ptr()


def err(a, b):
    """(max abs error, error relative to the reference's max magnitude)."""
    a, b = a.double(), b.double()
    d = (a - b).abs().max().item()
    return d, d / (b.abs().max().item() + 1e-30)


def assert_close(name, got, ref, rel, abs_floor=0.0):
    d, r = err(got, ref)
    assert torch.isfinite(got).all(), f'{name}: non-finite values'
    assert d <= rel * ref.abs().max().item() + abs_floor, \
        f'{name}: max|err|={d:.3e} (rel-to-max {r:.3e}) > tol {rel:.1e}*max + {abs_floor:.1e}'
    return d, r


def frag_to_dense(precision, wfrag, cout, cin):
    """Inverse of the fragment-order filter image (csrc/common.hpp: wfrag_index) -> [cout][9][cin] float32 CPU."""
    es = 2 if precision == 'bf16' else 4
    epu, kch = 16 // es, 32 // es
    flat = wfrag.float().cpu().reshape(-1)
    co = torch.arange(cout)[:, None, None]
    tap = torch.arange(9)[None, :, None]
    c = torch.arange(cin)[None, None, :]
    rec = ((co // 32) * 9 + tap) * (cin // kch) + c // kch
    lane = (co % 32) + 32 * ((c % kch) // epu)
    idx = rec * (64 * epu) + lane * epu + c % epu
    return flat[idx]


def pack_w(precision, w_oihw, cin_pad):
    """Run bdn_pack_weights; returns (wf, wd) device tensors."""
    dt, td = DT[precision]
    co, ci = w_oihw.shape[:2]
    wdev = dev(w_oihw)
    wf = torch.empty(co, 9, cin_pad, dtype=td, device='cuda')
    wd = torch.empty(cin_pad, 9, co, dtype=td, device='cuda') if cin_pad % 32 == 0 else None   # dgrad image: Cin_pad % 32
    _lib.call('bdn_pack_weights', dt, wdev.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else None,
              co, ci, cin_pad, st())
    return wf, wd


def bn_table(G, C, seed=0):
    """Random but well-conditioned [G][4][C] BatchNorm table (mean, invstd, scale, shift) as float32 CPU."""
    r = np.random.default_rng(seed)
    mean = r.uniform(-0.5, 0.5, (G, C))
    inv = r.uniform(0.5, 2.0, (G, C))
    gamma = r.uniform(0.5, 1.5, (G, C)) * np.where(r.uniform(0, 1, (G, C)) < 0.15, -1, 1)
    beta = r.uniform(-0.3, 0.3, (G, C))
    scale = gamma * inv
    shift = beta - mean * scale
    return torch.from_numpy(np.stack([mean, inv, scale, shift], 1).astype(np.float32))   # [G,4,C]


def bnrelu_ref(precision, z_nchw, bn, ipg):
    """relu(z*scale+shift) per group, rounded to the storage type like every kernel does."""
    out = torch.empty_like(z_nchw)
    G = bn.shape[0]
    for g in range(G):
        s = slice(g * ipg, (g + 1) * ipg)
        out[s] = torch.relu(z_nchw[s] * bn[g, 2][None, :, None, None] + bn[g, 3][None, :, None, None])
    return rnd(precision, out)


def preact(z_nchw, bn, ipg):
    """scale*z + shift per statistic group in float64 (the argument of the ReLU whose mask the fused producers apply)."""
    out = torch.empty_like(z_nchw, dtype=torch.float64)
    for g in range(bn.shape[0]):
        s = slice(g * ipg, (g + 1) * ipg)
        out[s] = z_nchw[s].double() * bn[g, 2].double()[None, :, None, None] + bn[g, 3].double()[None, :, None, None]
    return out


def assert_masked(name, got_nchw, full_nchw, pre, eps=1e-4):
    """The fused producers store the MASKED gradient g = dA * [scale z + shift > 0] (include/bidate_hip.h, bdn_conv3x3_dgrad_bs): `got` must
    equal `full` bit for bit where the ReLU is clearly on, be exactly zero where it is clearly off, and be one of the two within `eps`
    of the switching point (the device evaluates the pre-activation with one FMA in float32)."""
    got, full = got_nchw.double(), full_nchw.double()
    on, off = pre > eps, pre < -eps
    assert torch.equal(got[on], full[on]), f'{name}: stored gradient differs where the ReLU is on'
    assert (got[off] == 0).all(), f'{name}: stored gradient not zero where the ReLU is off'
    edge = ~(on | off)
    assert ((got[edge] == full[edge]) | (got[edge] == 0)).all(), f'{name}: stored gradient at the switching point is neither g nor 0'
    assert on.any() and off.any(), f'{name}: degenerate mask'
