import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from fabric_amd import _lib
lib=_lib.load(); st=_lib.stream_ptr(); dt=1; td=torch.bfloat16
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e3
torch.manual_seed(0)
shapes=[(128,64,64,128,128,1),(128,32,32,256,256,1),(128,16,16,512,512,1),(128,128,128,64,64,1),(64,128,128,128,64,0),(64,64,64,256,128,0),(6,24,40,64,64,1),(4,17,33,128,64,0)]
for (n,h,w,c,co,mode) in shapes:
    a0=torch.randn(n,h,w,c,device='cuda').to(td); dz=torch.randn(n,h,w,co,device='cuda').to(td)
    bn=torch.rand(2,4,c,device='cuda')+0.5; bn[:,3]-=1.0; ipg=n//2
    part=torch.empty(2*lib.bdn_wgrad_workspace_bytes(n,h,w,co,c,ipg)//4,device='cuda')
    res={}
    for v in (0,2):   # 0 = wgrad2, 2 = wgrad5 (producer / consumer)
        lib.bdn_set_tuning(2, v)
        dw=torch.zeros(co,c,3,3,device='cuda')
        args=(dt,dz.data_ptr(),co,a0.data_ptr(),c,None,0,mode,bn.data_ptr(),ipg,part.data_ptr(),dw.data_ptr(),c,n,h,w)
        t=timeit(lambda: _lib.call('bdn_conv3x3_wgrad',*args,st))
        t1=timeit(lambda: _lib.call('bdn_conv3x3_wgrad_ex',*args,1,st))
        _lib.call('bdn_conv3x3_wgrad',*args,st); torch.cuda.synchronize()
        res[v]=(dw.clone(),t,t1)
    d=(res[0][0]-res[2][0]).abs().max().item(); sc=res[0][0].abs().max().item()
    print(f'N={n} {h}x{w} {c}->{co} mode={mode}: v2 {res[0][1]:7.1f}/{res[0][2]:7.1f} us   v4 {res[2][1]:7.1f}/{res[2][2]:7.1f} us   maxdiff {d:.3e} (scale {sc:.3e})')
