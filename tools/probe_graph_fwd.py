"""Is a PARTIAL hipGraph capture of the training step -- the forward only: ~62 launches on one stream, no events, no second queue -- neutral on
the GPU, and what does it save on the host?  (test infrastructure)     python tools/probe_graph_fwd.py
Prints, for the B = 64 13-band 128x128 training forward: eager vs replayed-graph GPU time per forward and host time to enqueue it; then the same
forward inside a whole step (forward replayed from the graph, loss + backward + SGD eager) against the all-eager step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabric_amd import BiDateNet, _lib
from fabric_amd.train_step import TrainStep
B = 64
torch.manual_seed(0)
model = BiDateNet(13, 2).cuda().train()
ts = TrainStep(model, lr=1e-3)
x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
torch.cuda.set_stream(ts.stream())
eng, P = model.engine(), ts._state()


def timed(fn, n=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, host


for _ in range(3): ts.step(x1, x2, lbl)
fwd = lambda: eng.forward(x1, x2, P, training=True)
t_e, h_e = timed(fwd)
print(f'forward eager : {t_e:.3f} ms GPU-bound wall per forward, host enqueue {h_e:.3f} ms')
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=ts.stream()):
    logits, ws = eng.forward(x1, x2, P, training=True)
torch.cuda.synchronize()
t_g, h_g = timed(g.replay)
print(f'forward graph : {t_g:.3f} ms GPU-bound wall per forward, host enqueue {h_g:.3f} ms   ({(t_g / t_e - 1) * 100:+.1f} % GPU, {h_g - h_e:+.3f} ms host)')

# whole step with the forward replayed: the rest of TrainStep._step on the captured logits / workspace
tvn = _lib.load().bdn_overlap_workspace_bytes(B, 2, 128, 128, 0) // 4
tvws, loss, counts = torch.empty(tvn, device='cuda'), torch.empty((), device='cuda'), torch.empty(4, dtype=torch.int32, device='cuda')
dlogits = torch.empty_like(logits)


def step_graph_fwd():
    eng.invalidate_weights()
    g.replay()                                              # (the capture re-packs the filter images itself: they were stale when it was taken)
    st = _lib.stream_ptr()
    _lib.call('bdn_tversky', logits.data_ptr(), lbl.data_ptr(), 0.5, 0.5, 1e-7, tvws.data_ptr(), loss.data_ptr(), counts.data_ptr(), dlogits.data_ptr(), B, 2, 128, 128, st)
    eng.backward(ws, dlogits, P, ts.grads, on_ready=ts.bucketer.on_ready, zero_bias_grads=False)
    ts.bucketer.finish()
    _lib.call('bdn_sgd_step', ts.flat_params.data_ptr(), ts.flat_grads.data_ptr(), 1e-3, 1.0, ts.layout.total, st)


for rep in range(2):
    t_s, h_s = timed(lambda: ts.step(x1, x2, lbl))
    t_sg, h_sg = timed(step_graph_fwd)
    print(f'step eager {t_s:.3f} ms (host {h_s:.3f})   step with the forward from a graph {t_sg:.3f} ms (host {h_sg:.3f})   {(t_sg / t_s - 1) * 100:+.2f} %')
