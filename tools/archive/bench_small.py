"""Micro-benchmark (test infrastructure) of the HBM-bound stage kernels at the B=64 128x128 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib
dt, td = _lib.BDN_BF16, torch.bfloat16
st = _lib.stream_ptr(); lib = _lib.load()
B, S = 64, 128
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def rnd(*s): return torch.randn(*s, device='cuda').to(td)
z = rnd(B, S, S, 64); bn = torch.rand(2, 4, 64, device='cuda') + 0.5
w = torch.randn(2, 64, device='cuda'); b = torch.zeros(2, device='cuda')
logits = torch.empty(B, 2, S, S, device='cuda'); dl = torch.randn(B, 2, S, S, device='cuda')
dA = torch.empty_like(z); dw = torch.empty(2, 64, device='cuda'); db = torch.empty(2, device='cuda')
gb = lambda *ts: sum(t.numel() * t.element_size() for t in ts) / 1e9
t = timeit(lambda: _lib.call('bdn_outc_fwd', dt, z.data_ptr(), bn.data_ptr(), w.data_ptr(), b.data_ptr(), logits.data_ptr(), B, S, S, 64, 2, st))
print(f'outc_fwd {t:7.1f} us  {gb(z, logits) / t * 1e6:6.0f} GB/s')
ows = torch.empty(_lib.load().bdn_outc_bwd_workspace_bytes(dt, B, S, S, 64, 2) // 4, device='cuda')
t = timeit(lambda: _lib.call('bdn_outc_bwd', dt, dl.data_ptr(), z.data_ptr(), bn.data_ptr(), w.data_ptr(), dA.data_ptr(), dw.data_ptr(), db.data_ptr(), None, ows.data_ptr(), B, S, S, 64, 2, st))
print(f'outc_bwd {t:7.1f} us  {gb(z, dl, dA) / t * 1e6:6.0f} GB/s')
lbl = (torch.rand(B, S, S, device='cuda') < 0.1).to(torch.uint8)
ws = torch.empty(_lib.load().bdn_overlap_workspace_bytes(B, 2, S, S, 0) // 4, device='cuda'); loss = torch.empty(1, device='cuda'); cnt = torch.empty(4, dtype=torch.int32, device='cuda')
t = timeit(lambda: _lib.call('bdn_tversky', logits.data_ptr(), lbl.data_ptr(), 0.1, 0.9, 1e-7, ws.data_ptr(), loss.data_ptr(), cnt.data_ptr(), dl.data_ptr(), B, 2, S, S, st))
print(f'tversky  {t:7.1f} us  {gb(logits, logits, dl, lbl, lbl) / t * 1e6:6.0f} GB/s')
x1 = torch.randn(B, 13, S, S, device='cuda'); x0 = torch.empty(2 * B, S, S, 16, device='cuda', dtype=td)
t = timeit(lambda: _lib.call('bdn_pack_input', dt, x1.data_ptr(), x1.data_ptr(), x0.data_ptr(), B, 13, S, S, 16, st))
print(f'pack_in  {t:7.1f} us  {gb(x1, x1, x0) / t * 1e6:6.0f} GB/s')
# encoder level-1 sized kernels
z2 = rnd(2 * B, S, S, 64); dF = rnd(B, S, S, 128); dP = rnd(2 * B, S // 2, S // 2, 64); dA2 = torch.empty_like(z2)
t = timeit(lambda: _lib.call('bdn_enc_skip_bwd', dt, dF.data_ptr(), 128, z2.data_ptr(), bn.data_ptr(), dP.data_ptr(), dA2.data_ptr(), None, B, S, S, 64, st))
print(f'enc_skip_bwd L1 {t:7.1f} us  {(gb(z2, dP, dA2) + gb(dF) / 2) / t * 1e6:6.0f} GB/s')
f = torch.empty(B, S, S, 64, device='cuda', dtype=td)
t = timeit(lambda: _lib.call('bdn_fuse_product', dt, z2.data_ptr(), bn.data_ptr(), f.data_ptr(), B, S, S, 64, st))
print(f'fuse_product L1 {t:7.1f} us  {gb(z2, f) / t * 1e6:6.0f} GB/s')
pl = torch.empty(2 * B, S // 2, S // 2, 64, device='cuda', dtype=td)
t = timeit(lambda: _lib.call('bdn_bnrelu_pool', dt, z2.data_ptr(), bn.data_ptr(), B, pl.data_ptr(), 2 * B, S, S, 64, st))
print(f'pool L1         {t:7.1f} us  {gb(z2, pl) / t * 1e6:6.0f} GB/s')
src = rnd(B, S // 2, S // 2, 64); U = torch.empty(B, S, S, 64, device='cuda', dtype=td)
t = timeit(lambda: _lib.call('bdn_upsample2x', dt, src.data_ptr(), 1, bn.data_ptr(), U.data_ptr(), B, S // 2, S // 2, S, S, 64, st))
print(f'upsample L1     {t:7.1f} us  {gb(src, U) / t * 1e6:6.0f} GB/s')
dsrc = torch.empty_like(src)
t = timeit(lambda: _lib.call('bdn_upsample2x_bwd', dt, dF.data_ptr() + 64 * 2, 128, dsrc.data_ptr(), B, S // 2, S // 2, S, S, 64, st))
print(f'upsample_bwd L1 {t:7.1f} us  {(gb(dF) / 2 + gb(dsrc)) / t * 1e6:6.0f} GB/s')
n = 2 * B
wsb = torch.empty(lib.bdn_bn_bwd_workspace_bytes(dt, n, S, S, 64, B) // 4, device='cuda'); sums = torch.empty(2, 2, 64, device='cuda')
dg = torch.empty(64, device='cuda'); dbb = torch.empty(64, device='cuda'); dz = torch.empty_like(z2)
t = timeit(lambda: _lib.call('bdn_bn_bwd', dt, dA2.data_ptr(), 64, z2.data_ptr(), bn.data_ptr(), B, n, S, S, 64, wsb.data_ptr(), sums.data_ptr(), dg.data_ptr(), dbb.data_ptr(), dz.data_ptr(), st))
print(f'bn_bwd L1 (3 k) {t:7.1f} us  {gb(z2, z2, dA2, dA2, dz) / t * 1e6:6.0f} GB/s')
