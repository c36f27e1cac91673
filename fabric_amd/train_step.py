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
        """guard: when the step issues collectives (world > 1, or force_collectives) and guard_collectives() has not been called, the
        first step() runs it in its measure-only form (replace_streams=False: it may defer the buckets, it never swaps a stream the
        caller may already have adopted) and reports / warns about a stream arrangement in which they slow the step down."""
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
        compute units first (A/B tools/archive/ab_prio.py: -0.8 % step time).  The caller's current stream is joined on both
        sides, so the usual stream semantics hold for inputs and outputs."""
        if self._guard and self.collectives_report is None and self.bucketer.active():
            # measure only: the caller may already run its loop on step.stream(), which must not be swapped under it
            self.guard_collectives(*[int(v) for v in (x_d1.shape[0], x_d1.shape[2], x_d1.shape[3])], replace_streams=False)
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

    def _time_exchange(self, n):
        """Seconds per step of the bucket all-reduces ALONE: issued back to back from the chain's stream on an otherwise idle device
        (max over ranks).  The gradients are overwritten by the next backward anyway."""
        import time
        dev = self.flat_params.device
        b = self.bucketer
        with torch.cuda.stream(self.stream(dev)):
            for it in range(n + 2):
                if it == 2:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                works = [dist.all_reduce(b.flat[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a, e, _ in b.buckets]
                for w in works:
                    w.wait()
            torch.cuda.synchronize(dev)
            t = torch.tensor([(time.perf_counter() - t0) / n], dtype=torch.float64, device=dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def guard_collectives(self, B, H, W, steps=8, threshold=0.05, verbose=False, replace_streams=True):
        """Is the step slowed down by WHERE its collectives run?  RCCL's collective stream is created by torch, not by this library,
        and its hardware-queue placement relative to the chain / weight-gradient streams depends on creation order, priority and
        GPU_MAX_HW_QUEUES: one combination measured +48...+59 % step time, invisible to a sleep-kernel probe.
        So it is MEASURED, on synthetic inputs of the run's shape, `steps` steps each:

          local      the step without its collectives,
          overlap    the bucket all-reduces launched from backward (the intended arrangement),
          exchange   (only if overlap costs more than `threshold`) the five bucket all-reduces ALONE on an otherwise idle device: what
                     the exchange itself costs (xGMI transfer, RCCL's kernels) when nothing hides it.

        overlap <= local + 1.1 exchange + 2 % means the overhead is explained by the exchange even if none of it were hidden: the
        arrangement is fine, nothing is changed and nothing is warned about, however large it is (a slow interconnect is not a stream
        problem).  Anything beyond that is interference, i.e. a placement problem, and the remedies are tried in order -- (1) a new
        weight-gradient stream, (2) a new chain stream, each measured in the overlap arrangement, then (3) the buckets launched from
        the chain's stream after backward (bucketer.defer) on the best streams found.  (Deferring is a remedy, not a reference: with
        the collective stream on the chain's hardware queue it measured as slow as overlapping.)  Afterwards the arrangement that
        measured BEST is the one restored -- streams.restore() puts a displaced stream back -- and the report's `overhead_frac` is
        the number measured on exactly that arrangement.  With several ranks every decision is taken on the MAX
        over ranks, so all ranks walk the same path.  Parameters, BatchNorm buffers and the bucketer state are restored.

        replace_streams=False (what the automatic call inside the first step() uses): the stream remedies are skipped -- a caller that
        already runs its loop on step.stream() must not have that stream swapped under it -- and only deferring is available.
        Call this method yourself BEFORE `with torch.cuda.stream(step.stream())` to get the full repair (train.py and bench.py do)."""
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
        ok = False
        eng = self.model.engine()
        saved = saved_flat = None
        tried = []
        from . import streams as _streams
        orig_streams = (_streams.get('chain', dev), _streams.get('wgrad', dev))     # put back if a measurement raises half-way
        try:
            g = torch.Generator(device='cpu').manual_seed(99)
            C = self.model.n_channels
            x1 = torch.randn(B, C, H, W, generator=g).to(dev)
            x2 = (x1 + 0.3 * torch.randn(B, C, H, W, generator=g).to(dev))
            lbl = (torch.rand(B, H, W, generator=g) < 0.1).to(torch.uint8).to(dev)
            saved = {k: v.clone() for k, v in self._P.items()}
            saved_flat = self.flat_params.clone()

            def timed(collectives, defer=False):
                self.bucketer.enabled, self.bucketer.defer = collectives, defer
                # inputs and the saved copies were produced on the caller's stream: the chain stream (possibly a new one) joins it first
                self.stream(dev).wait_stream(torch.cuda.current_stream(dev))
                t = torch.tensor([self._time_steps(x1, x2, lbl, steps, 3)], dtype=torch.float64, device=dev)
                if self.world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                return float(t[0])

            def measure(name):
                t_local = timed(False)
                t_ov = timed(True)
                rec = {'arrangement': name, 'local_ms': t_local * 1e3, 'overlap_ms': t_ov * 1e3, 'overhead_frac': t_ov / t_local - 1.0,
                       'streams': (streams.get('chain', dev), streams.get('wgrad', dev))}
                tried.append(rec)
                if verbose:
                    print(f'guard_collectives: {name}: overlap overhead {rec["overhead_frac"] * 100:+.1f} %', flush=True)
                return rec

            first = measure('original')
            kept, placement, t_ex = first, None, None
            if first['overhead_frac'] > threshold:
                t_ex = self._time_exchange(steps)
                first['exchange_alone_ms'] = t_ex * 1e3
                explained = first['local_ms'] * 1.02 + 1.1 * t_ex * 1e3
                placement = first['overlap_ms'] > explained               # more than the fully exposed exchange would cost: interference
                if placement and replace_streams:
                    for remedy in ('new_wgrad_stream', 'new_chain_stream'):
                        if kept['overhead_frac'] <= threshold:
                            break
                        streams.replace('wgrad' if remedy == 'new_wgrad_stream' else 'chain', dev)
                        rec = measure(remedy)
                        if rec['overhead_frac'] < kept['overhead_frac']:
                            kept = rec
                # back to the best overlap arrangement found; if that is still a placement problem, try deferring ON it
                streams.restore('chain', kept['streams'][0], dev)
                streams.restore('wgrad', kept['streams'][1], dev)
                if placement and kept['overlap_ms'] > explained:
                    t_def = timed(True, defer=True)
                    kept['deferred_ms'] = t_def * 1e3
                    if t_def * 1e3 < kept['overlap_ms']:
                        kept = dict(kept, arrangement=kept['arrangement'] + ' + deferred_buckets', deferred=True,
                                    overhead_frac=t_def * 1e3 / kept['local_ms'] - 1.0)
            self.bucketer.defer = bool(kept.get('deferred'))
            rep = {'active': True, 'threshold': threshold, 'steps': steps, 'world': self.world,
                   'tried': [{k: v for k, v in r.items() if k != 'streams'} for r in tried],
                   'kept': kept['arrangement'], 'overhead_frac': kept['overhead_frac'], 'deferred_buckets': bool(kept.get('deferred')),
                   'placement_problem': placement, 'stream_remedies_allowed': bool(replace_streams),
                   'recovered': bool(placement) and kept is not first}
            still_bad = bool(placement) and kept['overhead_frac'] > threshold and \
                (kept['deferred_ms'] if kept.get('deferred') else kept['overlap_ms']) > kept['local_ms'] * 1.02 + 1.1 * t_ex * 1e3
            rep['ok'] = not still_bad
            rep['exchange_alone_ms'] = None if t_ex is None else t_ex * 1e3
            if still_bad:
                hint = ('' if replace_streams else '  The stream remedies were skipped because the guard ran inside step(): call '
                        'step.guard_collectives(B, H, W) before adopting step.stream().')
                warnings.warn(f'fabric_amd: launching the gradient all-reduces from backward costs the step {first["overhead_frac"] * 100:+.0f} % '
                              f'here, more than the exchange alone ({t_ex * 1e3:.2f} ms) explains -- a stream-placement problem; kept: {kept["arrangement"]} '
                              f'({kept["overhead_frac"] * 100:+.0f} %, threshold {threshold * 100:.0f} %).{hint}  Creating the process group BEFORE the '
                              f'first TrainStep, the default collective-stream priority and the default GPU_MAX_HW_QUEUES avoid it.', RuntimeWarning)
            ok = True
        finally:
            self.bucketer.enabled = True
            if not ok:
                self.bucketer.defer = False
                # a measurement raised (OOM, RCCL error) while a remedy's stream was in place: back to the arrangement the guard started
                # from -- the next guard must not start from a half-tried one (displaced streams are retired by restore, not leaked)
                _streams.restore('chain', orig_streams[0], dev)
                _streams.restore('wgrad', orig_streams[1], dev)
            self.bucketer.reset()
            torch.cuda.synchronize(dev)                    # chain-stream steps may still be in flight (exception path): restore after them
            if saved is not None and saved_flat is not None:
                for k, v in saved.items():
                    self._P[k].copy_(v)
                self.flat_params.copy_(saved_flat)
            eng.invalidate_weights()
            torch.cuda.synchronize(dev)
            if not ok:
                self.collectives_report = None             # a failed guard has measured nothing: the next call starts over
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
