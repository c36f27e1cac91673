"""Phase anatomy of one conv3x3 launch (apply tools/experimental/conv_stamp.patch, then tools/archive/build_variants.sh stamp "-DSTAMP"): every wave writes its
cycle-counter stamps into its tile's statistics row.  BIDATE_LIB=fabric_amd/csrc/variants/lib_stamp.so python tools/stamp_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib
lib = _lib.load(); st = _lib.stream_ptr(); dt = _lib.BDN_BF16
names = ['index math', 'loads issued', 'patch in LDS', 'barrier', 'main loop', 'acc -> LDS', 'barrier', 'copy-out']
for tag, n, h, w, c0, co, mode in (('e1b fwd', 128, 128, 128, 64, 64, 1), ('e2b fwd', 128, 64, 64, 128, 128, 1), ('e4b fwd', 128, 16, 16, 512, 512, 1),
                                   ('d4a-dgrad-like fwd', 64, 128, 128, 64, 128, 0), ('e1a fwd', 128, 128, 128, 16, 64, 0)):
    a0 = torch.randn(n, h, w, c0, device='cuda').to(torch.bfloat16)
    wt = (torch.randn(co, 9, c0, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(n, h, w, co, device='cuda', dtype=torch.bfloat16)
    bn = torch.rand(2, 4, c0, device='cuda') + 0.5
    bias = torch.zeros(co, device='cuda')
    nt = lib.bdn_conv3x3_num_mtiles(n, h, w, co, n // 2)
    stats = torch.zeros(nt, 2, co, device='cuda')
    fn = lambda: _lib.call('bdn_conv3x3', dt, a0.data_ptr(), c0, None, 0, mode, bn.data_ptr(), n // 2, wt.data_ptr(), bias.data_ptr(), out.data_ptr(),
                           stats.data_ptr(), n, h, w, co, st)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    s = stats[:, 0, :16].double()                      # wave 0 of column tile 0
    d = torch.cat([s[:, :1], s[:, 1:8] - s[:, 0:7]], 1).mean(0)
    tot = s[:, 7].mean().item()
    print(f'{tag}: {e0.elapsed_time(e1) * 1e3:.1f} us, {nt} tiles; wave lifetime {tot:.0f} clk (100 MHz ticks x?)')
    for nm, v in zip(names, d.tolist()):
        print(f'    {nm:14s} {v:9.0f}  {100 * v / tot:5.1f} %')
