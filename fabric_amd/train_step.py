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
                 process_group=None, n_buckets=4, distributed=True, force_collectives=False):
        self.model, self.lr = model, lr
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
