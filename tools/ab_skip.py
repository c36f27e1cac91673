"""Step-time sensitivity (test infrastructure): which launches is the step actually waiting for?  One process, one TrainStep; for every
named class of launches the step is timed with those launches DROPPED (fabric_amd._lib.SKIP; results are wrong, buffers keep the
previous step's plausible values), interleaved with the full step.  The time a class gives back when it disappears is an UPPER bound on
what any fusion / speed-up of it can return.      python tools/ab_skip.py [class ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep

def hw(args, i): return args[i]
# bdn_bn_bwd_apply(dtype, dA, ldA, z, bn, ipg, N, H, W, C, ...): H at 7; bdn_outc_bn_bwd_apply(..., B, H, W, C, ncls, st): H at -5
CLASSES = {
    'none': lambda n, a: False,
    'bn_bwd_apply@128,64': lambda n, a: (n == 'bdn_bn_bwd_apply' and a[7] >= 64) or n == 'bdn_outc_bn_bwd_apply',
    'bn_bwd_apply@all': lambda n, a: n in ('bdn_bn_bwd_apply', 'bdn_outc_bn_bwd_apply'),
    'finalize(fwd)': lambda n, a: n == 'bdn_bn_finalize',
    'finalize(bwd)': lambda n, a: n == 'bdn_bn_bwd_finalize',
    'upsample2x': lambda n, a: n == 'bdn_upsample2x',
    'upsample2x_bwd': lambda n, a: n in ('bdn_upsample2x_bwd', 'bdn_upsample2x_bwd_bs'),
    'product_pool': lambda n, a: n in ('bdn_product_pool', 'bdn_fuse_product'),
    'enc_skip_bwd': lambda n, a: n == 'bdn_enc_skip_bwd',
    'pack': lambda n, a: n in ('bdn_pack_input', 'bdn_pack_weights_multi'),
    'head+loss': lambda n, a: n in ('bdn_outc_fwd', 'bdn_tversky', 'bdn_outc_bwd'),
    'wgrad(all)': lambda n, a: n in ('bdn_conv3x3_wgrad_ex', 'bdn_conv3x3_wgrad_bnbwd'),
    'wgrad(decoder)': lambda n, a: n == 'bdn_conv3x3_wgrad_ex' and a[13] == 64,          # N = B: decoder layers
    'dgrad(all)': lambda n, a: n == 'bdn_conv3x3_dgrad_bs' or (n == 'bdn_conv3x3' and a[9] is None),     # no bias: data-gradient launches
    'conv fwd(all)': lambda n, a: n == 'bdn_conv3x3' and a[9] is not None,
    'all HBM-bound of the chain': lambda n, a: n not in ('bdn_conv3x3', 'bdn_conv3x3_dgrad_bs', 'bdn_conv3x3_wgrad_ex', 'bdn_conv3x3_wgrad_bnbwd', 'bdn_sgd_step'),
}
want = sys.argv[1:] or list(CLASSES)
B = 64
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3)
res = {k: [] for k in want}
with torch.cuda.stream(step.stream()):
    for _ in range(10): step.step(x1, x2, lbl)
    _lib.SKIP = lambda n, a: n == 'bdn_sgd_step'
    for rep in range(3):
        for k in want:
            _lib.SKIP = (lambda f: (lambda n, a: n == 'bdn_sgd_step' or f(n, a)))(CLASSES[k])      # never update the weights: a dropped launch leaves garbage gradients
            for _ in range(3): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 20)
            _lib.SKIP = lambda n, a: n == 'bdn_sgd_step'
            for _ in range(2): step.step(x1, x2, lbl)          # refill the buffers with real values
base = statistics.median(res['none']) if 'none' in res else None
for k in want:
    m = statistics.median(res[k])
    print(f'{k:32s} {m:7.3f} ms' + (f'  ({(m - base) * 1e3:+7.0f} us, {(m / base - 1) * 100:+5.1f} %)' if base else '') + f'   {[round(t, 3) for t in res[k]]}')
print('weights finite:', all(bool(torch.isfinite(p).all()) for p in model.parameters()))
