"""Time bdn_enc_skip_bwd on the five encoder levels of the benchmark shape (test infrastructure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib
B = 64
lib = _lib.load(); st = _lib.stream_ptr()
F32 = os.environ.get('DT') == 'f32'
dt, td, es = (_lib.BDN_F32, torch.float32, 4) if F32 else (_lib.BDN_BF16, torch.bfloat16, 2)
tot = 0
for (h, c) in ((128, 64), (64, 128), (32, 256), (16, 512), (8, 512)):
    z = torch.randn(2 * B, h, h, c, device='cuda').to(td)
    dF = torch.randn(B, h, h, 2 * c, device='cuda').to(td)
    dP = torch.randn(2 * B, h // 2, h // 2, c, device='cuda').to(td) if h > 8 else None
    bn = torch.rand(2, 4, c, device='cuda') + 0.5
    dA = torch.empty_like(z)
    rows = lib.bdn_enc_skip_bwd_rows(dt, B, h, h, c)
    part = torch.empty(2, rows, 2, c, device='cuda')
    fn = lambda: _lib.call('bdn_enc_skip_bwd', dt, dF.data_ptr(), 2 * c, z.data_ptr(), bn.data_ptr(), dP.data_ptr() if dP is not None else None,
                           dA.data_ptr(), part.data_ptr(), B, h, h, c, st)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    nbytes = (z.numel() * 2 + dF.numel() // 2 + (dP.numel() if dP is not None else 0)) * es
    tot += t
    print(f'{h:4d}x{h:<4d} C={c:4d}  {t:7.1f} us  {nbytes / t / 1e6:.2f} TB/s')
print('total', round(tot, 1), 'us')
