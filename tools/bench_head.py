"""Micro-benchmark (test infrastructure): the classifier / loss / SGD / packing launches of one B=64 128x128 bf16 step, each alone on
the chip through the C ABI, HIP events over 20 calls.  python tools/bench_head.py   (BIDATE_LIB selects a library variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib

B, H, W, C, NC = int(os.environ.get('B', 64)), 128, 128, 64, 2
dt, td = _lib.BDN_BF16, torch.bfloat16
lib = _lib.load()
st = _lib.stream_ptr()
dev = 'cuda'
torch.manual_seed(0)
z = torch.randn(B, H, W, C, device=dev).to(td)
bn = torch.rand(1, 4, C, device=dev) + 0.5
w = torch.randn(NC, C, device=dev) * 0.1
b = torch.zeros(NC, device=dev)
logits = torch.empty(B, NC, H, W, device=dev)
labels = (torch.rand(B, H, W, device=dev) < 0.1).to(torch.uint8)
dlogits = torch.empty_like(logits)
ws_t = torch.empty(lib.bdn_overlap_workspace_bytes(B, NC, H, W, 0) // 4 + 16, device=dev)
loss = torch.empty(1, device=dev); counts = torch.empty(4, dtype=torch.int32, device=dev)
rows = lib.bdn_outc_bwd_rows(dt, B, H, W, C)
stats = torch.empty(rows * 2 * C, device=dev)
ows = torch.empty(lib.bdn_outc_bwd_workspace_bytes(dt, B, H, W, C, NC) // 4, device=dev)
dw = torch.empty(NC, C, device=dev); db = torch.empty(NC, device=dev)
sums = torch.rand(1, 2, C, device=dev)
dz = torch.empty_like(z)
n_par = 13_401_154
p = torch.randn(n_par, device=dev); g = torch.randn(n_par, device=dev) * 1e-3


def timeit(name, fn, nbytes, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f'{name:24s} {us:8.1f} us   {nbytes / us * 1e-6:6.2f} TB/s')


P = lambda t: t.data_ptr()
if True:
    zb = z.numel() * 2
    timeit('outc_fwd', lambda: _lib.call('bdn_outc_fwd', dt, P(z), P(bn), P(w), P(b), P(logits), B, H, W, C, NC, st), zb + logits.numel() * 4)
    timeit('tversky (3 launches)', lambda: _lib.call('bdn_tversky', P(logits), P(labels), 0.1, 0.9, 1e-7, P(ws_t), P(loss), P(counts), P(dlogits), B, NC, H, W, st),
           2 * logits.numel() * 4 + 2 * labels.numel() + dlogits.numel() * 4)
    timeit('outc_bwd (+dw reduce)', lambda: _lib.call('bdn_outc_bwd', dt, P(dlogits), P(z), P(bn), P(w), None, P(dw), P(db), P(stats), P(ows), B, H, W, C, NC, st),
           zb + dlogits.numel() * 4)
    timeit('outc_bn_bwd_apply', lambda: _lib.call('bdn_outc_bn_bwd_apply', dt, P(dlogits), P(w), P(z), P(bn), B, P(sums), P(dz), B, H, W, C, NC, st),
           2 * zb + dlogits.numel() * 4)
    timeit('sgd', lambda: _lib.call('bdn_sgd_step', P(p), P(g), 1e-3, 1.0, n_par, st), 3 * n_par * 4)
