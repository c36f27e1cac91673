"""Micro-benchmark (test infrastructure): the HBM-bound launches of one BiDateNet(13,2) B=64 128x128 bf16 backward, each ALONE on the chip
through the C ABI -- encoder skip / unpool backward (5 levels), upsample backward (4), the first layer's weight gradient with BatchNorm
backward on load, BatchNorm backward apply (the 14 shapes of a step), product + pool (4).  Rates are ALGORITHMIC bytes (one read of every
input, one write of every output) over the HIP-event time of 20 calls.     python tools/bench_hbm.py [B=64]      (BIDATE_LIB selects a variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 128
dt, td, es = _lib.BDN_BF16, torch.bfloat16, 2
lib = _lib.load()
st = _lib.stream_ptr()
ENC = (64, 128, 256, 512, 512)
DEC_OUT = (256, 128, 64, 64)
P = lambda t: None if t is None else t.data_ptr()
tot = {}


def timeit(cls, name, fn, nbytes, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    a = tot.setdefault(cls, [0.0, 0.0]); a[0] += us; a[1] += nbytes
    print(f'{name:44s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us * 1e-6:6.2f} TB/s')


r = lambda *s: torch.randn(*s, device='cuda').to(td)
bnt = lambda g, c: torch.rand(g, 4, c, device='cuda') + 0.5

# ---- encoder skip + unpool backward
for k in range(1, 6):
    h = S >> (k - 1); c = ENC[k - 1]
    ldF = c if k == 5 else c + (ENC[4] if k == 4 else DEC_OUT[3 - k])       # dF is the skip half of the decoder's dcat (k = 5: dF5 alone)
    dF, z, bn = r(B, h, h, ldF), r(2 * B, h, h, c), bnt(2, c)
    dP = r(2 * B, h // 2, h // 2, c) if k < 5 else None
    dA = torch.empty(2 * B, h, h, c, device='cuda', dtype=td)
    rows = lib.bdn_enc_skip_bwd_rows(dt, B, h, h, c)
    bs = torch.empty(2 * rows * 2 * c, device='cuda')
    nb = (B * h * h * c + 2 * 2 * B * h * h * c + (dP.numel() if dP is not None else 0)) * es
    timeit('enc_skip_bwd', f'enc_skip_bwd level {k} ({h}x{h}x{c})',
           lambda: _lib.call('bdn_enc_skip_bwd', dt, P(dF), ldF, P(z), P(bn), P(dP), P(dA), P(bs), B, h, h, c, st), nb)
    del dF, z, dP, dA

# ---- upsample backward (with the previous decoder stage's BatchNorm-backward sums where the step fuses them)
cprev = ENC[4]
for j in range(1, 5):
    k = 5 - j
    H = S >> (k - 1); hs = H // 2
    ck = ENC[k - 1]
    dc = r(B, H, H, ck + cprev)
    dsrc = torch.empty(B, hs, hs, cprev, device='cuda', dtype=td)
    rows = lib.bdn_upsample2x_bwd_rows(dt, B, hs, hs, cprev) if j > 1 else 0
    nb = (B * H * H * cprev + B * hs * hs * cprev * (2 if rows else 1)) * es
    if rows:
        zp, bnp, bs = r(B, hs, hs, cprev), bnt(1, cprev), torch.empty(rows * 2 * cprev, device='cuda')
        fn = lambda: _lib.call('bdn_upsample2x_bwd_bs', dt, dc.data_ptr() + ck * es, ck + cprev, P(dsrc), P(zp), P(bnp), P(bs), B, hs, hs, H, H, cprev, st)
    else:
        fn = lambda: _lib.call('bdn_upsample2x_bwd', dt, dc.data_ptr() + ck * es, ck + cprev, P(dsrc), B, hs, hs, H, H, cprev, st)
    timeit('upsample2x_bwd', f'upsample2x_bwd{"_bs" if rows else ""} j={j} ({hs}->{H}, {cprev} ch of {ck + cprev})', fn, nb)
    cprev = DEC_OUT[j - 1]
    del dc, dsrc

# ---- the first layer's weight gradient (BatchNorm backward on load)
n = 2 * B
if lib.bdn_conv3x3_wgrad_bnbwd_supported(dt, n, S, S, 64, 16, B):
    dA, z, bn, sums, x0 = r(n, S, S, 64), r(n, S, S, 64), bnt(2, 64), torch.rand(2, 2, 64, device='cuda'), r(n, S, S, 16)
    part = torch.empty(lib.bdn_wgrad_workspace_bytes_ex(dt, n, S, S, 64, 16, 0, B, 0, _lib.wg_flags(3, 0, 256)) // 4, device='cuda')
    dw = torch.empty(64, 13, 3, 3, device='cuda')
    timeit('wgrad_first', 'wgrad_first (bn backward on load)',
           lambda: _lib.call('bdn_conv3x3_wgrad_bnbwd', dt, P(dA), 64, P(z), P(bn), P(sums), B, 64, P(x0), 16, P(part), P(dw), 13, n, S, S, st),
           (2 * n * S * S * 64 + n * S * S * 16) * es)
    del dA, z, x0

# ---- BatchNorm backward apply: the 14 launches of a step (e1b and d4a are folded into their data-gradient convs, d4b is outc_bn_bwd_apply, e1a is wgrad_first)
shapes = []
for k in range(1, 6):
    h = S >> (k - 1)
    for nm in ('a', 'b'):
        if (k, nm) not in ((1, 'a'), (1, 'b')):
            shapes.append((f'e{k}{nm}', 2 * B, h, ENC[k - 1], B))
for j in range(1, 5):
    h = S >> (4 - j)
    for nm in ('a', 'b'):
        if (j, nm) not in ((4, 'a'), (4, 'b')):
            shapes.append((f'd{j}{nm}', B, h, DEC_OUT[j - 1], B))
for name, n, h, c, ipg in shapes:
    dA, z, bn = r(n, h, h, c), r(n, h, h, c), bnt(n // ipg, c)
    G = n // ipg
    rows = 8
    part = torch.rand(G * rows * 2 * c, device='cuda')
    sums = torch.empty(G, 2, c, device='cuda'); dg, db = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    dz = torch.empty_like(dA)
    scr = torch.empty(max(lib.bdn_bn_bwd_scratch_bytes(G, c) // 8, 1), dtype=torch.float64, device='cuda')
    timeit('bn_bwd_apply', f'bn_bwd_apply {name} ({n}x{h}x{h}x{c})',
           lambda: _lib.call('bdn_bn_bwd_apply', dt, P(dA), c, P(z), P(bn), ipg, n, h, h, c, P(part), rows, 1, P(sums), P(dg), P(db), P(dz), P(scr), st),
           3 * n * h * h * c * es)
    del dA, z, dz

# ---- product + pool (forward)
for k in range(1, 5):
    h = S >> (k - 1); c = ENC[k - 1]
    z, bn = r(2 * B, h, h, c), bnt(2, c)
    f, pool = torch.empty(B, h, h, c, device='cuda', dtype=td), torch.empty(2 * B, h // 2, h // 2, c, device='cuda', dtype=td)
    timeit('product_pool', f'product_pool level {k} ({h}x{h}x{c})',
           lambda: _lib.call('bdn_product_pool', dt, P(z), P(bn), P(f), P(pool), B, h, h, c, st), (z.numel() + f.numel() + pool.numel()) * es)
    del z, f, pool

for k, (us, nb) in tot.items():
    print(f'sum {k:16s} {us:8.1f} us  {nb / 1e6:8.1f} MB  {nb / us * 1e-6:6.2f} TB/s')
