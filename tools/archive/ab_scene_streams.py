"""Scene inference on one vs two streams (test infrastructure):  python tools/archive/ab_scene_streams.py [size] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.utils import inference as inf
size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().eval()
g = torch.Generator(device='cuda').manual_seed(3)
d1 = torch.randn(13, size, size, device='cuda', generator=g)
d2 = d1 + 0.3 * torch.randn(13, size, size, device='cuda', generator=g)
n = len(inf.tile_origins(size, size, 128)[0])
masks = {}
for batch in ([int(sys.argv[2])] if len(sys.argv) > 2 else [128, 256]):
    for two in (False, True, False, True):
        inf.predict_scene(model, d1, d2, 128, batch, two_streams=two); torch.cuda.synchronize()
        t = time.perf_counter()
        m = inf.predict_scene(model, d1, d2, 128, batch, two_streams=two); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        masks[two] = m
        print(f'batch {batch} two_streams={two}: {dt:.4f} s  {n / dt:.0f} tiles/s')
    print('masks equal:', bool(torch.equal(masks[False], masks[True])))
