"""Run one conv3x3 / wgrad shape a few times (for rocprofv3 --pmc runs).  args: kind N H W C0 C1 Cout mode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import _lib
kind = sys.argv[1]; n, h, w, c0, c1, co, mode = (int(v) for v in sys.argv[2:9])
dt, td = _lib.BDN_BF16, torch.bfloat16
lib = _lib.load(); st = _lib.stream_ptr()
a0 = torch.randn(n, h, w, c0, device='cuda').to(td)
a1 = torch.randn(n, h, w, c1, device='cuda').to(td) if c1 else None
bn = torch.rand(2, 4, c0, device='cuda') + 0.5
ipg = n // 2
if kind == 'conv':
    wt = (torch.randn(co, 9, c0 + c1, device='cuda') * 0.05).to(td)
    out = torch.empty(n, h, w, co, device='cuda', dtype=td)
    bias = torch.zeros(co, device='cuda')
    stats = torch.empty(lib.bdn_conv3x3_num_mtiles(n, h, w, co, ipg) * 2 * co, device='cuda')
    for _ in range(3):
        _lib.call('bdn_conv3x3', dt, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1, mode, bn.data_ptr(), ipg,
                  wt.data_ptr(), bias.data_ptr(), out.data_ptr(), stats.data_ptr(), n, h, w, co, st)
else:
    dz = torch.randn(n, h, w, co, device='cuda').to(td)
    part = torch.empty(lib.bdn_wgrad_workspace_bytes(n, h, w, co, c0 + c1, ipg) // 4, device='cuda')
    dw = torch.empty(co, c0 + c1, 3, 3, device='cuda')
    for _ in range(3):
        _lib.call('bdn_conv3x3_wgrad', dt, dz.data_ptr(), co, a0.data_ptr(), c0, a1.data_ptr() if c1 else None, c1,
                  mode, bn.data_ptr(), ipg, part.data_ptr(), dw.data_ptr(), c0 + c1, n, h, w, st)
torch.cuda.synchronize()
