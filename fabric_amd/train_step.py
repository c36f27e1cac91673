"""The train.py step (reference train.py:83-101 + optim.SGD, train.py:55) as one fused schedule:

    to-device tensors -> forward -> Tversky loss (+ argmax TP/FP/FN counts) -> backward
    -> bucketed gradient all-reduce overlapped with backward -> SGD

with no host synchronisation inside the step (the reference syncs every step for sklearn
P/R/F1 and a comet upload, train.py:103-115; here the counts stay on the device).
Parameters and gradients live in two flat float32 buffers (fabric_amd/parallel.py); the
module's nn.Parameters are re-pointed at views of them, so ``model.state_dict()``,
``model.parameters()`` and ``p.grad`` keep working for callers that expect the reference's
surface.
"""
import torch
import torch.distributed as dist

from . import _lib
from .engine import param_order
from .parallel import FlatLayout, GradBucketer


class TrainStep:
    def __init__(self, model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9, eps=1e-7,
                 process_group=None, n_buckets=4, distributed=True, force_collectives=False, guard=True):
        """guard: when the step issues collectives (world > 1, or force_collectives), the first step() first runs
        guard_collectives() -- a few timed steps with and without the bucket all-reduces -- and repairs / reports a stream
        arrangement in which they slow the step down (see there)."""
        self.model, self.lr = model, lr
        self._guard = guard
        self.collectives_report = None
        self.alpha, self.beta, self.eps = tversky_alpha, tversky_beta, eps
        self.group = process_group
        self.high_priority_chain = True
        self._hp = None
        self.world = (dist.get_world_size(process_group)
                      if distributed and dist.is_available() and dist.is_initialized() else 1)
        named = list(model.named_parameters())
        dev = named[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError('fabric_amd: TrainStep needs the model on a ROCm device (model.cuda() first)')
        order = param_order(model.n_channels)
        self.layout = FlatLayout([(k, p.shape) for k, p in named], order)
        self.flat_params = torch.zeros(self.layout.total, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(self.layout.total, dtype=torch.float32, device=dev)
        for k, p in named:
            v = self.layout.view(self.flat_params, k)
            v.copy_(p.data)
            p.data = v
            p.grad = self.layout.view(self.flat_grads, k)
        self.grads = {k: self.layout.view(self.flat_grads, k) for k, _ in named}
        bias_tail = [k for k in order if k.endswith('.bias') and k.split('.')[-2] in ('0', '3')]
        self.bucketer = GradBucketer(self.layout, self.flat_grads, n_buckets, process_group, keys_no_reduce=bias_tail,
                                     enabled=self.world > 1 or force_collectives, force=force_collectives)
        if self.world > 1:                                   # identical start on every rank (DataParallel broadcasts)
            dist.broadcast(self.flat_params, src=0, group=process_group)
        self._tv = None
        self.last_counts = None
        # detached aliases of every parameter / buffer (same storage), built once: no per-step dict walk
        self._P = {k: v.detach() for k, v in self.model.state_dict(keep_vars=True).items()}

    def _state(self):
        return self._P

    def step(self, x_d1, x_d2, labels):
        """One optimisation step.  Returns the loss as a fresh 0-dim device tensor (no sync; safe to keep in a list like
        the reference loop does with `cd_loss`).  `last_counts` / `last_logits` are persistent buffers that the NEXT step
        overwrites: clone them if they have to outlive it.

        The step's dependency chain (forward, loss, dz chain, SGD) is enqueued on a HIGH-priority HIP stream; the
        weight-gradient GEMMs run beside it on a normal-priority stream (engine.backward), so the chain's kernels get
        compute units first (A/B tools/ab_prio.py: -0.8 % step time).  The caller's current stream is joined on both
        sides, so the usual stream semantics hold for inputs and outputs."""
        if self._guard and self.collectives_report is None and self.bucketer.active():
            self.guard_collectives(*[int(v) for v in (x_d1.shape[0], x_d1.shape[2], x_d1.shape[3])])
        if not self.high_priority_chain:
            return self._step(x_d1, x_d2, labels)
        cur = torch.cuda.current_stream(x_d1.device)
        hp = self.stream(x_d1.device)
        if cur.cuda_stream == hp.cuda_stream:              # the caller already runs its loop on the step's stream: no joins
            return self._step(x_d1, x_d2, labels)
        self._hp.wait_stream(cur)
        with torch.cuda.stream(self._hp):
            loss = self._step(x_d1, x_d2, labels)
        cur.wait_stream(self._hp)
        for t in (loss, self.last_logits, self.last_counts):
            t.record_stream(cur)
        return loss

    def stream(self, device=None):
        """The high-priority stream the step's chain runs on: the process-wide 'chain' stream of the device (fabric_amd/streams.py;
        every TrainStep shares it, so the N-th instance of a process is as fast as the first).  A training loop that makes it
        the current stream (``with torch.cuda.stream(step.stream()): ...``) saves the two cross-stream joins per step
        (~25 us of idle GPU)."""
        from . import streams
        self._hp = streams.get('chain', device if device is not None else self.flat_params.device)      # not cached: streams.replace() may swap it
        return self._hp

    # ------------------------------------------------------------------ run-time guard for the collectives' stream placement
    def _time_steps(self, x1, x2, lbl, n, warm):
        import time
        dev = x1.device
        with torch.cuda.stream(self.stream(dev)):
            for _ in range(warm):
                self._step(x1, x2, lbl)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                self._step(x1, x2, lbl)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / n

    def guard_collectives(self, B, H, W, steps=8, threshold=0.05, verbose=False):
        """Is the step slowed down by WHERE its collectives run?  RCCL's collective stream is created by torch, not by this library,
        and its hardware-queue placement relative to the chain / weight-gradient streams depends on creation order, priority and
        GPU_MAX_HW_QUEUES: one combination measured +48...+59 % step time, invisible to a sleep-kernel probe.
        So it is MEASURED: `steps` steps on synthetic inputs of the run's shape with the bucket all-reduces and without; while the
        overhead exceeds `threshold` the remedies are tried in order and kept only if they help --
          1. a new weight-gradient stream (streams.replace: the old one is parked so that its queue slot stays taken),
          2. a new chain stream,
          3. the buckets launched from the CHAIN's stream at the end of backward (bucketer.defer: no overlap with backward any more,
             but no interference either -- costs the exposed transfer instead of half a step);
        and a RuntimeWarning says so when none of them brings it under the threshold.  With several ranks the decision is taken on the
        MAX over ranks, so every rank walks the same path.  Parameters, BatchNorm buffers and the bucketer state are restored:
        the model is exactly as before.  Returns (and keeps in `collectives_report`) what was measured."""
        import warnings
        from . import streams
        dev = self.flat_params.device
        if not self.bucketer.active():
            self.collectives_report = {'active': False}
            return self.collectives_report
        backend = str(dist.get_backend(self.group))
        if 'nccl' not in backend:                                       # gloo (tests on one GPU / CPU): reductions run on the host, no collective stream
            self.collectives_report = {'active': True, 'guarded': False, 'reason': f'backend {backend}: no device-side collective stream'}
            return self.collectives_report
        self.collectives_report = {'running': True}                     # re-entrancy: _step below must not call the guard again
        g = torch.Generator(device='cpu').manual_seed(99)
        C = self.model.n_channels
        x1 = torch.randn(B, C, H, W, generator=g).to(dev)
        x2 = (x1 + 0.3 * torch.randn(B, C, H, W, generator=g).to(dev))
        lbl = (torch.rand(B, H, W, generator=g) < 0.1).to(torch.uint8).to(dev)
        saved = {k: v.clone() for k, v in self._P.items()}
        saved_flat = self.flat_params.clone()
        eng = self.model.engine()

        def overhead():
            self.bucketer.enabled = False
            t_local = self._time_steps(x1, x2, lbl, steps, 3)
            self.bucketer.enabled = True
            t_coll = self._time_steps(x1, x2, lbl, steps, 3)
            t = torch.tensor([t_local, t_coll], dtype=torch.float64, device=dev)
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            t_local, t_coll = float(t[0]), float(t[1])
            return t_coll / t_local - 1.0, t_local, t_coll

        tried = []
        try:
            ov, tl, tc = overhead()
            tried.append({'remedy': 'none', 'overhead_frac': ov, 'local_ms': tl * 1e3, 'collectives_ms': tc * 1e3})
            best = ov
            for remedy in ('new_wgrad_stream', 'new_chain_stream', 'deferred_buckets'):
                if best <= threshold:
                    break
                if remedy == 'new_wgrad_stream':
                    streams.replace('wgrad', dev)
                elif remedy == 'new_chain_stream':
                    streams.replace('chain', dev)
                else:
                    self.bucketer.defer = True
                ov, tl, tc = overhead()
                tried.append({'remedy': remedy, 'overhead_frac': ov, 'local_ms': tl * 1e3, 'collectives_ms': tc * 1e3})
                if remedy == 'deferred_buckets' and ov >= best:
                    self.bucketer.defer = False                        # did not help: keep the overlapped launches
                best = min(best, ov)
                if verbose:
                    print(f'guard_collectives: {remedy}: overhead {ov * 100:+.1f} %', flush=True)
        finally:
            self.bucketer.enabled = True
            self.bucketer.reset()
            for k, v in saved.items():
                self._P[k].copy_(v)
            self.flat_params.copy_(saved_flat)
            eng.invalidate_weights()
            torch.cuda.synchronize(dev)
        rep = {'active': True, 'threshold': threshold, 'steps': steps, 'world': self.world, 'tried': tried,
               'overhead_frac': tried[-1]['overhead_frac'] if tried else None, 'deferred_buckets': bool(self.bucketer.defer),
               'recovered': len(tried) > 1 and best <= threshold, 'ok': best <= threshold}
        if best > threshold:
            warnings.warn(f'fabric_amd: the gradient all-reduces cost the step {best * 100:+.0f} % (threshold {threshold * 100:.0f} %) and no '
                          f'stream re-arrangement brought that down: {tried}.  Try creating the process group BEFORE the first TrainStep, the '
                          f'default (not high) collective-stream priority, and the default GPU_MAX_HW_QUEUES.', RuntimeWarning)
        self.collectives_report = rep
        return rep

    def _step(self, x_d1, x_d2, labels):
        model = self.model
        eng = model.engine()
        P = self._state()
        logits, ws = eng.forward(x_d1, x_d2, P, training=True)
        B, C, H, W = logits.shape
        dev = logits.device
        if self._tv is None or self._tv[3] != (B, C, H, W):
            n = _lib.load().bdn_overlap_workspace_bytes(B, C, H, W, 0) // 4
            self._tv = (torch.empty(n, dtype=torch.float32, device=dev), torch.empty((), dtype=torch.float32, device=dev),
                        torch.empty(4, dtype=torch.int32, device=dev), (B, C, H, W))
        tvws, loss, counts, _ = self._tv
        if labels.dtype != torch.uint8:
            labels = labels.to(torch.uint8)
        labels = labels.contiguous()
        dlogits = torch.empty_like(logits)
        st = _lib.stream_ptr()
        _lib.call('bdn_tversky', logits.data_ptr(), labels.data_ptr(), float(self.alpha), float(self.beta),
                  float(self.eps), tvws.data_ptr(), loss.data_ptr(), counts.data_ptr(), dlogits.data_ptr(), B, C, H, W, st)
        eng.backward(ws, dlogits, P, self.grads, on_ready=self.bucketer.on_ready, zero_bias_grads=False)
        self.bucketer.finish()
        # p -= lr * (sum of rank gradients) / world : per-rank loss, averaged gradients (standard DDP)
        _lib.call('bdn_sgd_step', self.flat_params.data_ptr(), self.flat_grads.data_ptr(), float(self.lr),
                  1.0 / self.world, self.layout.total, st)
        eng.invalidate_weights()                              # packed bf16/f32 GEMM images are now stale
        self.last_counts = counts
        self.last_logits = logits
        return loss.clone()
