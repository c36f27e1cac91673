"""Host -> device input path of the training loop (reference train.py:83-85: `batch_img1.to(dev)` etc., fed by the
DataLoader of utils/helpers.py:250-257).

The reference copies every float32 NCHW batch synchronously on the compute stream.  At the benchmark shape that is
2 x 54.5 MB + 1 MB of labels per step, ~1.8 ms over PCIe Gen5 x16 -- a quarter of a training step if it is not hidden.
`DeviceFeeder` hides it: a dedicated copy stream moves batch k+1 from pinned host memory into one of `depth` (3) device slots
while the step of batch k runs; events order slot reuse (a slot is refilled only after the step that read it has finished)
and hand-off (the step waits for its slot's copies only).  Batches that arrive in pageable memory are first staged into
pinned buffers by a few host threads (a single-threaded 109 MB memcpy would take longer than the step).

    feeder = DeviceFeeder(device)
    for x1, x2, y in feeder(loader):            # device tensors; valid until the next iteration
        loss = step.step(x1, x2, y)
"""
from concurrent.futures import ThreadPoolExecutor

import torch


class _Slot:
    def __init__(self):
        self.dev = None          # device tensors of this slot
        self.pin = None          # pinned staging tensors (only for pageable sources)
        self.ready = torch.cuda.Event()
        self.free = None         # recorded on the consumer's stream after it used the slot
        self.issued = False


class DeviceFeeder:
    def __init__(self, device, depth=3, stage_threads=4):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('fabric_amd: DeviceFeeder needs a ROCm device')
        self.device = torch.device(device)
        self.depth = max(2, depth)       # 3: a slot is refilled two steps after it was read (2 slots: +7.5 % step time, 3: +2.5 %)
        self.slots = [_Slot() for _ in range(self.depth)]
        self.pool = ThreadPoolExecutor(max_workers=stage_threads) if stage_threads > 1 else None
        self.stage_threads = stage_threads

    @property
    def copy_stream(self):
        """The process-wide copy stream of the device, fetched on every use (streams.replace() may have swapped it)."""
        from . import streams
        return streams.get('copy', self.device)

    def close(self):
        """Stop the staging threads and drop the device slots / pinned buffers (the copy stream is process-wide and stays)."""
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        self.slots = [_Slot() for _ in range(self.depth)]

    # ------------------------------------------------------------------ host side
    def _pinned(self, slot, batch):
        """The batch in pinned memory: as it is when the loader already pinned it, else copied into the slot's staging buffers
        by `stage_threads` threads (torch releases the GIL inside copy_)."""
        if all(t.is_pinned() for t in batch):
            return batch
        if slot.pin is None or any(p.shape != t.shape or p.dtype != t.dtype for p, t in zip(slot.pin, batch)):
            slot.pin = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in batch]
        elif slot.issued:
            slot.ready.synchronize()      # the DMA that read this staging buffer `depth` batches ago must be over before the host
                                          # overwrites it (the host runs far ahead of the copy stream, which waits for the consumer)
        jobs = []
        for p, t in zip(slot.pin, batch):
            t = t.contiguous()
            n = t.shape[0] if t.dim() else 1
            parts = min(self.stage_threads, n) if self.pool and t.numel() * t.element_size() > (4 << 20) else 1
            if parts <= 1:
                p.copy_(t)
                continue
            step = (n + parts - 1) // parts
            jobs += [self.pool.submit(p[i:i + step].copy_, t[i:i + step]) for i in range(0, n, step)]
        for j in jobs:
            j.result()
        return slot.pin

    def _issue(self, slot, batch):
        batch = [torch.as_tensor(t) for t in batch]
        src = self._pinned(slot, batch)
        cs = self.copy_stream
        if slot.free is not None:
            cs.wait_event(slot.free)                     # the step that read this slot last is done with it
        with torch.cuda.stream(cs):
            if slot.dev is None or any(d.shape != t.shape or d.dtype != t.dtype for d, t in zip(slot.dev, src)):
                # allocated UNDER the copy stream: a block handed out by the consumer stream's pool may still be in use by kernels
                # that stream has queued (its temporaries are freed on the host long before they are dead on the device), and
                # the copy stream would write into it unordered
                slot.dev = [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in src]
            for d, t in zip(slot.dev, src):
                d.copy_(t, non_blocking=True)
            slot.ready.record(cs)
        slot.issued = True

    # ------------------------------------------------------------------ iteration
    def __call__(self, batches):
        """Yields device batches; the copy of the following batch is already in flight when a batch is handed out."""
        it = iter(batches)
        try:
            nxt = next(it)
        except StopIteration:
            return
        k = 0
        self._issue(self.slots[0], nxt)
        while True:
            slot = self.slots[k % self.depth]
            try:
                nxt = next(it)
                self._issue(self.slots[(k + 1) % self.depth], nxt)
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot.ready)
            for t in slot.dev:
                t.record_stream(cur)                     # allocated on the copy stream, read on the consumer's
            yield tuple(slot.dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))   # whatever the consumer enqueued on its stream reads the slot before this
            slot.free = ev
            if nxt is None:
                return
            k += 1
