"""3x3x3 convolution throughput on the layer shapes of a multi-date 3-D U-Net with BiDateNet's widths (BASELINE configs[3]:
5 dates x 13 bands x 128 x 128; pooling over H, W only).  python tools/bench_conv3d.py [samples=8]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd.conv3d import Conv3d3x3

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = 5
layers = [(13, 64, 128), (64, 64, 128), (64, 128, 64), (128, 128, 64), (128, 256, 32), (256, 256, 32), (256, 512, 16), (512, 512, 16),
          (512, 256, 32), (256, 128, 64), (128, 64, 128)]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot = {'fwd': [0, 0], 'dgrad': [0, 0], 'wgrad': [0, 0]}
for ci, co, s in layers:
    op = Conv3d3x3(torch.randn(co, ci, 3, 3, 3, device='cuda') * 0.05, torch.zeros(co, device='cuda'))
    x = torch.randn(N, D, s, s, op.cp, device='cuda').to(torch.bfloat16)
    dz = torch.randn(N, D, s, s, co, device='cuda').to(torch.bfloat16)
    fl = 2.0 * N * D * s * s * co * 27 * ci
    line = f'{ci:4d}->{co:4d} @ {D}x{s}x{s} '
    for what, fn in (('fwd', lambda: op.forward(x)), ('dgrad', (lambda: op.dgrad(dz)) if op.wd is not None else None), ('wgrad', lambda: op.wgrad(dz, x))):
        if fn is None:
            line += f'| {what} n/a '
            continue
        t = timeit(fn)
        tot[what][0] += t; tot[what][1] += fl
        line += f'| {what} {t * 1e6:7.1f} us {fl / t / 1e12:6.1f} TF/s '
    print(line)
print(json.dumps({'workload': f'3x3x3 conv layers of a 5-date 3-D U-Net, {N} samples of 5x13x128x128, bf16',
                  **{k: {'ms': round(v[0] * 1e3, 3), 'TFLOPs': round(v[1] / v[0] / 1e12, 1)} for k, v in tot.items()}}))
