"""The first conv's fused BatchNorm-backward + weight-gradient GEMM alone, against the two-kernel path (apply + simple wgrad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr(); dt = 1; td = torch.bfloat16
N, H, W, C, C0, ipg = 128, 128, 128, 64, 16, 64
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
dA = torch.randn(N, H, W, C, device='cuda').to(td); z = torch.randn(N, H, W, C, device='cuda').to(td)
x = torch.randn(N, H, W, C0, device='cuda').to(td); bn = torch.rand(2, 4, C, device='cuda') + 0.5
sums = torch.randn(2, 2, C, device='cuda'); dz = torch.empty_like(z)
part = torch.empty(lib.bdn_wgrad_workspace_bytes(N, H, W, C, C0, ipg) // 4, device='cuda'); dw = torch.empty(C, 13, 3, 3, device='cuda')
rows = 512
sp = torch.randn(2 * rows, 2, C, device='cuda'); dg = torch.empty(C, device='cuda'); db = torch.empty(C, device='cuda')
scr = torch.empty(lib.bdn_bn_bwd_scratch_bytes(2, C), dtype=torch.uint8, device='cuda')
t_fused = timeit(lambda: _lib.call('bdn_conv3x3_wgrad_bnbwd', dt, dA.data_ptr(), C, z.data_ptr(), bn.data_ptr(), sums.data_ptr(), ipg, C,
                                   x.data_ptr(), C0, part.data_ptr(), dw.data_ptr(), 13, N, H, W, st))
t_apply = timeit(lambda: _lib.call('bdn_bn_bwd_apply', dt, dA.data_ptr(), C, z.data_ptr(), bn.data_ptr(), ipg, N, H, W, C, sp.data_ptr(), rows, 1,
                                   sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dz.data_ptr(), scr.data_ptr(), st))
t_wg = timeit(lambda: _lib.call('bdn_conv3x3_wgrad', dt, dz.data_ptr(), C, x.data_ptr(), C0, None, 0, 0, None, ipg, part.data_ptr(), dw.data_ptr(), 13, N, H, W, st))
byt = N * H * W * (2 * C + C0) * 2
print(f'fused GEMM + reduce {t_fused:.1f} us ({byt / t_fused / 1e6:.2f} TB/s of dA + z + x)   two-kernel path: finalize + apply {t_apply:.1f} us + wgrad {t_wg:.1f} us')
