"""Micro-benchmark (test infrastructure): the bf16x3 convolutions of one BiDateNet(13,2) B=64 128x128 step -- forward (three terms) and data
gradient (two and three terms) per layer shape, through the C ABI with HIP events.   python tools/bench_x3_conv.py [B=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib
from fabric_amd.engine import build_layers

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 128
X3, X2 = _lib.BDN_BF16X3, _lib.BDN_BF16X2
dims = [(S >> k, S >> k) for k in range(5)]
st = _lib.stream_ptr()
lib = _lib.load()
_warm = [False]


def timeit(fn, iters=6):
    if not _warm[0]:
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        _warm[0] = True
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot = {}
for L in build_layers(13):
    h, w = dims[L.level - 1]
    n = 2 * B if L.enc else B
    ipg = B
    for what, dt, cin, cout, terms in (('fwd', X3, L.cin, L.cout, 3), ('dgrad2', X2, L.cout, L.cin, 2), ('dgrad3', X3, L.cout, L.cin, 3)):
        if what != 'fwd' and L.name == 'e1a':
            continue
        sp = torch.randn(n, h, w, 2 * cin, device='cuda').to(torch.bfloat16)
        wt = (torch.randn(cout, 9, 3 * cin, device='cuda') * 0.05).to(torch.bfloat16)
        out = torch.empty(n, h, w, cout, device='cuda')
        nt = lib.bdn_conv3x3_num_mtiles_ex(dt, n, h, w, cin, cout, ipg)
        stats = torch.empty(nt * 2 * cout, device='cuda') if what == 'fwd' else None
        fn = lambda: _lib.call('bdn_conv3x3', dt, sp.data_ptr(), cin, None, 0, 0, None, ipg, wt.data_ptr(), None, out.data_ptr(),
                               stats.data_ptr() if stats is not None else None, n, h, w, cout, st)
        t = timeit(fn)
        if what == 'fwd' and cin % 64 == 0 and cin <= 512 and L.name[2] == 'b':
            # the same stage from the float32 operand (BatchNorm+ReLU + split inside the staging, split operand left for the weight gradient)
            # against the split pass + convolution it replaces
            zf = torch.randn(n, h, w, cin, device='cuda')
            bn = torch.rand(n // ipg, 4, cin, device='cuda') + 0.5
            so = torch.empty(n, h, w, 2 * cin, device='cuda', dtype=torch.bfloat16)
            t_sp = timeit(lambda: _lib.call('bdn_split_pack', zf.data_ptr(), cin, None, 0, 1, bn.data_ptr(), ipg, so.data_ptr(), n, h, w, st))
            f_src = lambda keep: _lib.call('bdn_conv3x3_x3src', dt, zf.data_ptr(), cin, 1, bn.data_ptr(), ipg, wt.data_ptr(), None, out.data_ptr(),
                                           stats.data_ptr(), so.data_ptr() if keep else None, n, h, w, cout, st)
            t_src, t_src0 = timeit(lambda: f_src(True)), timeit(lambda: f_src(False))
            a = tot.setdefault('fwd_b_split+conv', [0.0, 0.0]); a[0] += t + t_sp; a[1] += 2.0 * n * h * w * cout * 9 * cin * terms
            a = tot.setdefault('fwd_b_x3src', [0.0, 0.0]); a[0] += t_src; a[1] += 2.0 * n * h * w * cout * 9 * cin * terms
            print(f'{L.name} split_pack {t_sp * 1e6:7.1f} us + conv {t * 1e6:7.1f} us = {(t + t_sp) * 1e6:7.1f} us   x3src {t_src * 1e6:7.1f} us (without the by-product {t_src0 * 1e6:7.1f} us)')
            del zf, bn, so
        fl = 2.0 * n * h * w * cout * 9 * cin * terms
        a = tot.setdefault(what, [0.0, 0.0]); a[0] += t; a[1] += fl
        print(f'{L.name} {what:7s} {n:4d}x{h:3d}x{w:3d} {cin:5d}->{cout:4d} {t * 1e6:8.1f} us {fl / t / 1e12:7.1f} TFLOP/s (executed bf16 MFMA work)')
        del sp, wt, out
for k, (t, f) in tot.items():
    print(f'sum {k}: {t * 1e3:.3f} ms  {f / t / 1e12:.1f} TFLOP/s')
