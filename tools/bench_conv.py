"""Micro-benchmark (test infrastructure): times every conv3x3 (fwd + dgrad) and wgrad launch shape of one
BiDateNet(13,2) B=64 128x128 bf16 step through the C ABI, with HIP events.  python tools/bench_conv.py [conv|wgrad|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib
from fabric_amd.engine import build_layers, ENC_CH

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
B, S = int(os.environ.get('B', 64)), 128
dt, td, es = _lib.BDN_BF16, torch.bfloat16, 2
dims = [(S >> k, S >> k) for k in range(5)]
st = _lib.stream_ptr()
lib = _lib.load()
shapes = []   # (tag, N, H, W, C0, C1, Cout, mode)
for L in build_layers(13):
    h, w = dims[L.level - 1]
    n = 2 * B if L.enc else B
    if L.name[2] == 'a' and not L.enc:
        ck = ENC_CH[L.level - 1]
        c0, c1 = ck, L.cin - ck
    else:
        c0, c1 = L.cin, 0
    mode = 1 if L.name[2] == 'b' else 0
    shapes.append((L.name + ' fwd', n, h, w, c0, c1, L.cout, mode))
    if L.name != 'e1a':
        shapes.append((L.name + ' dgrad', n, h, w, L.cout, 0, L.cin, 0))

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

tot_t = tot_f = 0
if what in ('conv', 'all'):
    for tag, n, h, w, c0, c1, co, mode in shapes:
        a0 = torch.randn(n, h, w, c0, device='cuda').to(td)
        a1 = torch.randn(n, h, w, c1, device='cuda').to(td) if c1 else None
        wt = (torch.randn(co, 9, c0 + c1, device='cuda') * 0.05).to(td)
        if os.environ.get('ZERO'):          # DVFS probe: all-zero operands draw less power -> higher clock (MI355X_MICROARCH.md)
            a0.zero_(); wt.zero_()
            if a1 is not None: a1.zero_()
        out = torch.empty(n, h, w, co, device='cuda', dtype=td)
        bn = torch.rand(2, 4, c0, device='cuda') + 0.5
        bias = torch.zeros(co, device='cuda')
        ipg = n // 2 if n == 2 * B else n
        nt = lib.bdn_conv3x3_num_mtiles(n, h, w, co, ipg)
        stats = torch.empty(nt * 2 * co, device='cuda')
        fwd = 'fwd' in tag
        if os.environ.get('BS') and tag[2:] == 'b dgrad':          # as in the training step: + BatchNorm-backward sums of the 'a' layer in the epilogue
            zprev = torch.randn(n, h, w, co, device='cuda').to(td)
            bnp = torch.rand(2, 4, co, device='cuda') + 0.5
            part = torch.empty(nt * 2 * co, device='cuda')
            fn = lambda: _lib.call('bdn_conv3x3_dgrad_bs', dt, a0.data_ptr(), c0, wt.data_ptr(), out.data_ptr(), zprev.data_ptr(), bnp.data_ptr(), ipg,
                                   part.data_ptr(), n, h, w, co, st)
            tag = tag + '+bs'
        else:
          fn = lambda: _lib.call('bdn_conv3x3', dt, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1, mode, bn.data_ptr(), ipg,
                               wt.data_ptr(), bias.data_ptr() if fwd else None, out.data_ptr(), stats.data_ptr() if fwd else None, n, h, w, co, st)
        t = timeit(fn)
        fl = 2.0 * n * h * w * co * 9 * (c0 + c1)
        tot_t += t; tot_f += fl
        print(f'{tag:13s} N={n:3d} {h:3d}x{w:3d} Cin={c0 + c1:4d} Cout={co:4d} mode={mode}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s')
    print(f'conv total {tot_t * 1e3:.3f} ms  {tot_f / tot_t / 1e12:.1f} TF/s')
wt_t = wt_f = 0
if what in ('wgrad', 'all'):
    for L in build_layers(13):
        h, w = dims[L.level - 1]
        n = 2 * B if L.enc else B
        ipg = B
        if L.name[2] == 'a' and not L.enc:
            ck = ENC_CH[L.level - 1]; c0, c1 = ck, L.cin - ck
        else:
            c0, c1 = L.cin, 0
        mode = 1 if L.name[2] == 'b' else 0
        a0 = torch.randn(n, h, w, c0, device='cuda').to(td)
        a1 = torch.randn(n, h, w, c1, device='cuda').to(td) if c1 else None
        dz = torch.randn(n, h, w, L.cout, device='cuda').to(td)
        bn = torch.rand(2, 4, c0, device='cuda') + 0.5
        part = torch.empty(lib.bdn_wgrad_workspace_bytes(n, h, w, L.cout, c0 + c1, ipg) // 4, device='cuda')
        dw = torch.empty(L.cout, L.cin_real, 3, 3, device='cuda')
        fl = 2.0 * n * h * w * L.cout * 9 * (c0 + c1)
        line = f'{L.name:5s} wgrad N={n:3d} {h:3d}x{w:3d} Cin={c0 + c1:4d} Cout={L.cout:4d} '
        # the kernel the training step runs for the shape (role-split where the shape class allows it, BatchNorm on load for the b layers)
        flg = _lib.wg_flags(1, 0, int(os.environ.get('WG_BLOCKS', 0)))
        ran = lib.bdn_conv3x3_wgrad_variant(dt, n, h, w, L.cout, c0, c1, ipg, mode, flg)
        t = timeit(lambda: _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), L.cout, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1,
                                     mode, bn.data_ptr(), ipg, part.data_ptr(), dw.data_ptr(), L.cin_real, n, h, w, flg, st))
        line += f' | kernel {ran} {t * 1e6:7.1f} us {fl / t / 1e12:6.1f} TF/s'
        wt_t += t; wt_f += fl
        t = timeit(lambda: _lib.call('bdn_conv3x3_wgrad_ex', dt, dz.data_ptr(), L.cout, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1,
                                     0, bn.data_ptr(), ipg, part.data_ptr(), dw.data_ptr(), L.cin_real, n, h, w, 2, st))
        line += f' | reduce {t * 1e6:6.1f} us (ws {part.numel() * 4 / 1e6:.0f} MB)'
        print(line)
    print(f'wgrad GEMMs on the default plan {wt_t * 1e3:.3f} ms  {wt_f / wt_t / 1e12:.1f} TF/s')
