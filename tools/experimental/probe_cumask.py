"""EXPERIMENT (needs tools/experimental/cumask_stream.hip.inc compiled into the library and bound in _lib.py).  CU-masked streams: (1) does a mask restrict a kernel (one wide conv on n CUs), (2) the training
step with the weight-gradient stream on `n` CUs and the chain on the rest / on all.  python tools/probe_cumask.py"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.call('bdn_stream_create_cumask', C.cast(words, C.c_void_p), 8, C.byref(out))
    return torch.cuda.ExternalStream(out.value, device=dev)


def low(n, start=0):
    return list(range(start, start + n))


# ---------------------------------------------------------------- (1) one conv under masks
lib = _lib.load()
n, h, w, c0, co = 128, 32, 32, 256, 256
a0 = torch.randn(n, h, w, c0, device=dev).to(torch.bfloat16)
wt = (torch.randn(co, 9, c0, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(n, h, w, co, device=dev, dtype=torch.bfloat16)
bn = torch.rand(2, 4, c0, device=dev) + 0.5
torch.cuda.synchronize()
for label, bits in (('all 256', low(256)), ('low 128', low(128)), ('high 128', low(128, 128)), ('low 64', low(64)), ('even 128', list(range(0, 256, 2))),
                    ('low 192', low(192))):
    s = masked_stream(bits)
    with torch.cuda.stream(s):
        fn = lambda: _lib.call('bdn_conv3x3', _lib.BDN_BF16, a0.data_ptr(), c0, None, 0, 0, bn.data_ptr(), n // 2, wt.data_ptr(), None, out.data_ptr(), None,
                               n, h, w, co, s.cuda_stream)
        fn(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): fn()
        e1.record(s); s.synchronize()
    print(f'conv e3b on {label:9s}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us', flush=True)

# ---------------------------------------------------------------- (2) the training step
B = 64
x1 = torch.randn(B, 13, 128, 128, device=dev); x2 = torch.randn(B, 13, 128, 128, device=dev)
lbl = (torch.rand(B, 128, 128, device=dev) < 0.1).to(torch.uint8)
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().train()
step = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
eng = model.engine()
hp0 = step.stream(dev)
side0 = eng._side_stream(dev)
key = str(dev)


def run(label, chain, side, blocks):
    step._hp = chain
    from fabric_amd import streams as _streams
    _streams._streams[(0, "wgrad")] = side          # fabric_amd/streams.py owns the step's streams since round 3
    eng.wgrad_blocks = blocks
    ts = []
    with torch.cuda.stream(chain):
        for rep in range(3):
            for _ in range(5): step.step(x1, x2, lbl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): step.step(x1, x2, lbl)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
    print(f'{label:44s} blocks={blocks:3d}: median {statistics.median(ts):.3f} ms  {[round(t, 3) for t in ts]}', flush=True)


run('default (hp chain, plain side)', hp0, side0, 0)
for nside in (64, 96, 128, 160):
    side = masked_stream(low(nside))
    rest = masked_stream(low(256 - nside, nside))
    for blocks in (0, 256):
        run(f'side {nside} CUs, chain on the other {256 - nside}', rest, side, blocks)
        run(f'side {nside} CUs, chain hp on all', hp0, side, blocks)
run('default again', hp0, side0, 0)
