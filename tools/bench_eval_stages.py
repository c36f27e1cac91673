"""Micro-benchmark (test infrastructure): every launch of the eval-shaped forward (fabric_amd/engine.py::_forward_eval) at the scene leg's
shapes -- B tile pairs of 13 x 128 x 128, bf16 -- one at a time through the C ABI, with HIP events.
    python tools/bench_eval_stages.py [B=256] [name-filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib
from fabric_amd.engine import build_layers, ENC_CH

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flt = sys.argv[2].split(',') if len(sys.argv) > 2 else []
S = 128
dt, td = _lib.BDN_BF16, torch.bfloat16
dims = [(S >> k, S >> k) for k in range(5)]
st = _lib.stream_ptr()


_warm = [False]


def timeit(fn, iters=10):
    if not _warm[0]:                      # the clock ramps for tens of ms after a pause (profiles/r5_operand_reuse.txt): warm up by whole launches
        import time
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(20): fn()
            torch.cuda.synchronize()
        _warm[0] = True
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot_t = tot_f = 0.0
for L in build_layers(13):
    if flt and not any(f in L.name for f in flt):
        continue
    h, w = dims[L.level - 1]
    enc_b = L.enc and L.name[2] == 'b'
    n = 2 * B if L.enc else B
    if L.name[2] == 'a' and not L.enc:
        c0 = ENC_CH[L.level - 1]; c1 = L.cin - c0
    else:
        c0, c1 = L.cin, 0
    a0 = torch.randn(n, h, w, c0, device='cuda').relu().to(td)
    a1 = torch.randn(n, h, w, c1, device='cuda').relu().to(td) if c1 else None
    wt = (torch.randn(L.cout, 9, c0 + c1, device='cuda') * (2.0 / (9 * (c0 + c1))) ** 0.5).to(td)
    sc, sh = torch.rand(L.cout, device='cuda') + 0.5, torch.randn(L.cout, device='cuda') * 0.1
    fl = 2.0 * n * h * w * L.cout * 9 * L.cin_real
    if enc_b:
        f = torch.empty(B, h, w, L.cout, device='cuda', dtype=td)
        pool = torch.empty(n, h // 2, w // 2, L.cout, device='cuda', dtype=td) if L.level < 5 else None
        fn = lambda: _lib.call('bdn_conv3x3_eval_pair', dt, a0.data_ptr(), c0, wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), f.data_ptr(),
                               pool.data_ptr() if pool is not None else None, B, h, w, L.cout, st)
        tag = 'pair' + ('+pool' if pool is not None else '')
    elif L.name == 'd4b':
        cw, cb = torch.randn(2, 64, device='cuda'), torch.zeros(2, device='cuda')
        mask = torch.empty(n, h, w, dtype=torch.uint8, device='cuda')
        fn = lambda: _lib.call('bdn_conv3x3_eval_cls', dt, a0.data_ptr(), c0, wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), None,
                               cw.data_ptr(), cb.data_ptr(), 2, None, mask.data_ptr(), None, 0, 0, n, h, w, L.cout, st)
        tag = 'cls->mask'
    else:
        out = torch.empty(n, h, w, L.cout, device='cuda', dtype=td)
        fn = lambda: _lib.call('bdn_conv3x3_eval', dt, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1, wt.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                               out.data_ptr(), None, None, n, h, w, L.cout, st)
        tag = ''
    t = timeit(fn)
    tot_t += t; tot_f += fl
    print(f'{L.name} {n:4d}x{h:3d}x{w:3d} {c0 + c1:5d}->{L.cout:4d} {tag:10s} {t * 1e6:8.1f} us {fl / t / 1e12:7.1f} TFLOP/s  {fl / t / 2.5e15:5.3f}')
    del a0, a1, wt
print(f'sum {tot_t * 1e3:.3f} ms  {tot_f / tot_t / 1e12:.1f} TFLOP/s = {tot_f / tot_t / 2.5e15:.3f} of peak')
