"""Time the weight-gradient split-K GEMM and its reduction separately (bdn_conv3x3_wgrad_ex phases 1 / 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib
from fabric_amd.engine import build_layers, ENC_CH
lib=_lib.load(); st=_lib.stream_ptr(); dt=1; td=torch.bfloat16
B,S=64,128
dims=[(S>>k,S>>k) for k in range(5)]
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e3
tg=tr=0
for L in build_layers(13):
    h,w=dims[L.level-1]; n=2*B if L.enc else B
    if L.name[2]=='a' and not L.enc:
        ck=ENC_CH[L.level-1]; c0,c1=ck,L.cin-ck
    else: c0,c1=L.cin,0
    mode=1 if L.name[2]=='b' else 0
    ipg=n//2 if n==2*B else n
    a0=torch.randn(n,h,w,c0,device='cuda').to(td); a1=torch.randn(n,h,w,c1,device='cuda').to(td) if c1 else None
    dz=torch.randn(n,h,w,L.cout,device='cuda').to(td); bn=torch.rand(2,4,c0,device='cuda')+0.5
    wsb=lib.bdn_wgrad_workspace_bytes(n,h,w,L.cout,c0+c1,ipg)
    part=torch.empty(wsb//4,device='cuda'); dw=torch.empty(L.cout,c0+c1,3,3,device='cuda')
    f=lambda ph: _lib.call('bdn_conv3x3_wgrad_ex',dt,dz.data_ptr(),L.cout,a0.data_ptr(),c0,a1.data_ptr() if c1 else None,c1,mode,bn.data_ptr(),ipg,part.data_ptr(),dw.data_ptr(),c0+c1,n,h,w,ph,st)
    g=timeit(lambda: f(1)); r=timeit(lambda: f(2)); tg+=g; tr+=r
    print(f'{L.name} Cin={c0+c1:4d} Cout={L.cout:4d}: gemm {g:7.1f} us  reduce {r:6.1f} us  ({wsb/1e6:5.1f} MB partials -> {wsb/r/1e6:5.2f} TB/s)')
print(f'total gemm {tg/1e3:.3f} ms  reduce {tr/1e3:.3f} ms')
