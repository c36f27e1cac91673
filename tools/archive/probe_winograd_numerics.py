"""Gate 0 of the reduced-MFMA-work probe: what does a Winograd convolution cost in LOGIT ERROR at bf16?

Runs on the CPU (test infrastructure: it drives the oracle, never the product).  The oracle's conv3x3 is replaced by emulations of
what a bf16 MFMA kernel computes:
  direct   operands rounded to bf16, products accumulated in float32, z stored as bf16 (the shipped kernels);
  wino1d   F(2,3) along the image width only: 4 multiplies per 2 outputs per filter row (1.5x fewer MACs).  V = B^T d and U = G g
           are formed in float32 and rounded to bf16 (they are the MFMA operands), M accumulates in float32 over channels AND
           the three filter rows, the output transform y0 = m0+m1+m2, y1 = m1-m2-m3 runs on the float32 accumulators;
  wino2d   F(2x2,3x3): 16 multiplies per 4 outputs (2.25x fewer MACs), same rounding points.
applied to the layers a reduced-work kernel would serve (Cin >= 128 and Cout >= 128: e2b..e5b, d1a..d2b).  Prints max / mean
|logit - float32 oracle| on the golden-vector inputs G2 (13 bands, 128x128, B = 2) and G6 (dates from different distributions);
tests/test_gpu_model.py's bf16 bound is max 0.25 / mean 0.03.

    python tools/archive/probe_winograd_numerics.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from oracle import bidate_oracle as O
from oracle import filler


def bf(t):
    return t.to(torch.bfloat16).float()


def direct(x, w, b=None):
    z = F.conv2d(bf(x), bf(w), None, padding=1)
    if b is not None:
        z = z + b[None, :, None, None]
    return bf(z)


def _wide(x, w):
    return w.shape[1] >= 128 and w.shape[0] >= 128 and x.shape[3] % 2 == 0 and x.shape[2] % 2 == 0


def wino1d(x, w, b=None):
    if not _wide(x, w):
        return direct(x, w, b)
    x = bf(x)
    xp = F.pad(x, (1, 1, 0, 0))                                  # zero padding along the width
    d0, d1, d2, d3 = xp[..., 0:-3:2], xp[..., 1:-2:2], xp[..., 2:-1:2], xp[..., 3::2]
    V = [bf(d0 - d2), bf(d1 + d2), bf(d2 - d1), bf(d1 - d3)]
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]                 # [co, ci, ky] per horizontal tap; U from the float32 master weights
    U = [bf(g0), bf((g0 + g1 + g2) * 0.5), bf((g0 - g1 + g2) * 0.5), bf(g2)]
    M = [F.conv2d(V[i], U[i].unsqueeze(-1), None, padding=(1, 0)) for i in range(4)]      # 3x1 kernels: rows stay direct
    y0, y1 = M[0] + M[1] + M[2], M[1] - M[2] - M[3]
    z = torch.stack([y0, y1], dim=-1).reshape(x.shape[0], w.shape[0], x.shape[2], x.shape[3])
    if b is not None:
        z = z + b[None, :, None, None]
    return bf(z)


def wino2d(x, w, b=None):
    if not _wide(x, w):
        return direct(x, w, b)
    x = bf(x)
    Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
    N, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                   # [N, C, H/2, W/2, 4, 4]
    V = bf(torch.einsum('ij,ncyxjk,lk->ncyxil', Bt, tiles, Bt))
    U = bf(torch.einsum('ij,ocjk,lk->ocil', G, w, G))
    M = torch.einsum('ncyxil,ocil->noyxil', V, U)
    Y = torch.einsum('ij,noyxjk,lk->noyxil', At, M, At)          # [N, O, H/2, W/2, 2, 2]
    z = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)
    if b is not None:
        z = z + b[None, :, None, None]
    return bf(z)


def run(conv, sd, x1, x2):
    keep = O.conv3x3
    O.conv3x3 = conv
    try:
        with torch.no_grad():
            return O.bidate_forward(sd, x1, x2, training=True)[0]
    finally:
        O.conv3x3 = keep


if __name__ == '__main__':
    torch.set_num_threads(8)
    from fabric_amd.models.bidate_model import BiDateNet
    cases = [('G2', dict(seed=0)), ('G6', dict(seed=0, different_dates=True)), ('seed 5', dict(seed=5))]
    for tag, kw in cases:
        model = filler.fill_module(BiDateNet(13, 2, precision='fp32'))
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        x1, x2, _ = filler.make_inputs(2, 13, 128, **kw)
        x1, x2 = torch.from_numpy(x1), torch.from_numpy(x2)
        ref = run(F_conv := (lambda x, w, b=None: F.conv2d(x, w, b, padding=1)), sd, x1, x2)
        print(f'{tag}: logit std {ref.std():.3f}')
        for name, fn in (('direct bf16', direct), ('wino1d F(2,3) bf16', wino1d), ('wino2d F(2x2,3x3) bf16', wino2d)):
            lg = run(fn, sd, x1, x2)
            d = (lg - ref).abs()
            agree = (lg.argmax(1) == ref.argmax(1)).float().mean()
            print(f'   {name:24s} max|dlogit| {d.max():.4f}  mean {d.mean():.5f}  argmax agreement {agree:.4f}')
