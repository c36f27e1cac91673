import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/fabric_amd') else os.environ['GRAFT_REPO_ROOT'])
import torch
from fabric_amd import _lib
st=_lib.stream_ptr()
for (B,h,w,C,ld) in [(64,64,64,64,128),(64,32,32,128,256),(64,16,16,256,512),(64,8,8,512,1024)]:
    H,W=2*h,2*w
    dU=torch.randn(B,H,W,ld,device='cuda').bfloat16(); out=torch.empty(B,h,w,C,device='cuda',dtype=torch.bfloat16)
    f=lambda: _lib.call('bdn_upsample2x_bwd', 1, dU.data_ptr()+ (ld-C)*2, ld, out.data_ptr(), B,h,w,H,W,C,st)
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/10*1e3
    mb=(B*H*W*C*2+B*h*w*C*2)/1e6
    print(f'ups_bwd {h}x{w} C={C}: {t:.1f} us  {mb/t*1e-3*1e3/1e3:.2f} TB/s eff ({mb:.0f} MB)')
    rows=_lib.load().bdn_upsample2x_bwd_rows(1,B,h,w,C)
    if rows:
        z=torch.randn(B,h,w,C,device='cuda').bfloat16(); bn=torch.rand(1,4,C,device='cuda')+0.5; part=torch.empty(rows,2,C,device='cuda')
        f=lambda: _lib.call('bdn_upsample2x_bwd_bs', 1, dU.data_ptr()+ (ld-C)*2, ld, out.data_ptr(), z.data_ptr(), bn.data_ptr(), part.data_ptr(), B,h,w,H,W,C,st)
        f(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(f'   with fused BatchNorm-backward sums: {e0.elapsed_time(e1)/10*1e3:.1f} us ({rows} rows)')
