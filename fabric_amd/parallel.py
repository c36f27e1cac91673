"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process nn.DataParallel wrap (utils/helpers.py:333-335:
per-forward parameter broadcast + scatter/gather) with the DDP pattern: every rank holds the
full model, patch-pair batches are sharded across ranks, BatchNorm statistics stay local
(exactly what DataParallel's replicas do), and the 53.6 MB of float32 gradients are summed
with a handful of bucketed all-reduces launched from inside backward, so the collectives run
on RCCL's stream while the remaining dgrad / wgrad kernels keep the compute stream busy.

The gradients live in ONE flat buffer laid out in the order they complete during backward
(engine.param_order), so a bucket is a contiguous slice and needs no packing copy.  xGMI is
point to point (7 links per GPU): a ring all-reduce is bound by one link, so buckets are kept
large (default 4 slices of ~13 MB, the last one split once more so that only ~1 MB is left for the end of backward)
rather than many small NVSwitch-style ones.

This module is pure torch.distributed logic and runs unchanged on CPU tensors with the gloo
backend (tests/test_host_cpu.py::test_grad_bucketer_gloo_world2, world_size 2); tests/test_gpu_ddp.py runs the full step with
two ranks on one GPU (gloo) and, where two devices are visible, over RCCL.
"""
import torch
import torch.distributed as dist
import torch.utils.data


def init_rccl(rank, world_size, device, high_priority=False):
    """torch.distributed over RCCL ('nccl' IS RCCL on ROCm) for one process per GPU.

    The collectives' internal stream keeps the DEFAULT priority.  A high-priority collective stream looks attractive (a bucket
    all-reduce is small and on the path to the optimizer step) but measured, with one rank and forced buckets, +48...+59 % step
    time when the group is created after the step's streams (+1 % when created first).  Mechanism (round 4):
    HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues per priority class in creation order, and in that arrangement
    RCCL's collective stream lands on the hardware queue of the step's high-priority chain stream, so every bucket all-reduce
    serialises with the dz chain; a NEW chain stream (created after the group) removes it (+0.6 %).  The library cannot see RCCL's
    stream, so TrainStep.guard_collectives MEASURES the overhead on the real arrangement before a loop starts and re-creates the
    step's streams when it is the slow one; bench.py reports what it saw (`collectives`)."""
    import os
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the host driver only supports dmabuf IPC
    kw = {}
    if high_priority:
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            kw['pg_options'] = opts
        except Exception:                                          # older / differently built torch: default priority
            kw = {}
    dist.init_process_group('nccl', rank=rank, world_size=world_size, device_id=torch.device(device), **kw)


class FlatLayout:
    """Offsets of every parameter inside the flat parameter / gradient buffers."""

    def __init__(self, named_shapes, order):
        shapes = dict(named_shapes)
        assert set(order) == set(shapes), 'param order must cover exactly the model parameters'
        self.order = list(order)
        self.slices = {}
        off = 0
        for k in self.order:
            n = 1
            for d in shapes[k]:
                n *= d
            n_pad = (n + 3) // 4 * 4                     # keep every tensor 16-byte aligned
            self.slices[k] = (off, n, tuple(shapes[k]))
            off += n_pad
        self.total = off

    def view(self, flat, key):
        off, n, shape = self.slices[key]
        return flat[off:off + n].view(shape)


class GradBucketer:
    """Launches an async all-reduce on a contiguous slice of the flat gradient buffer as soon as every
    gradient inside it has been produced.

    keys_no_reduce: gradients that are identically zero on every rank (conv biases in front of a
    BatchNorm) -- they sit at the tail of the layout and are never communicated."""

    def __init__(self, layout, flat_grads, n_buckets=4, group=None, keys_no_reduce=(), enabled=True, tail_bytes=1 << 20,
                 force=False):
        """n_buckets equal slices by bytes, plus one more cut in front of the last `tail_bytes` of gradients: the final
        bucket cannot start before the very last weight gradient of backward exists, so its all-reduce is the one piece of
        communication nothing can hide -- in BiDateNet the last megabyte is the four shallow encoder convs, while an equal
        quarter (13 MB) would also have held back three deep layers that were ready a millisecond earlier."""
        self.layout, self.flat, self.group = layout, flat_grads, group
        self.enabled = enabled                      # False: purely local step even inside an initialised process group
        self.force = force                          # True: issue the bucket all-reduces even in a world of one rank (a way to run the
                                                    # RCCL launches, their stream ordering and their cost on a single GPU)
        self.defer = False                          # True: no launches from inside backward; finish() issues every bucket from the
                                                    # caller's stream (TrainStep.guard_collectives' last remedy)
        skip = set(keys_no_reduce)
        keys = [k for k in layout.order if k not in skip]
        assert keys == layout.order[:len(keys)], 'non-reduced gradients must form the tail of the layout'
        end = 0
        for k in keys:
            off, n, _ = layout.slices[k]
            end = max(end, off + n)
        self.reduce_end = (end + 3) // 4 * 4
        target = max(1, self.reduce_end // max(1, n_buckets))
        self.buckets = []                                # (start, stop, [keys])
        cur, start = [], 0
        for k in keys:
            off, n, _ = layout.slices[k]
            cur.append(k)
            stop = (off + n + 3) // 4 * 4
            if stop - start >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append((start, stop, cur))
                cur, start = [], stop
        if cur:
            # split the last bucket in front of its final `tail_bytes` (whole tensors only; never an empty half)
            cut, acc = len(cur), 0
            for i in range(len(cur) - 1, 0, -1):
                acc += layout.slices[cur[i]][1] * 4
                if acc > tail_bytes:
                    break
                cut = i
            if tail_bytes > 0 and 0 < cut < len(cur):
                mid = layout.slices[cur[cut]][0]
                self.buckets.append((start, mid, cur[:cut]))
                self.buckets.append((mid, self.reduce_end, cur[cut:]))
            else:
                self.buckets.append((start, self.reduce_end, cur))
        self.key_bucket = {k: i for i, (_, _, ks) in enumerate(self.buckets) for k in ks}
        self.reset()

    def reset(self):
        self.pending = [set(ks) for _, _, ks in self.buckets]
        self.works = []
        self.launched = [False] * len(self.buckets)

    def world_size(self):
        if not self.enabled:
            return 1
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def active(self):
        """Do the buckets go through the process group?"""
        if not (self.enabled and dist.is_available() and dist.is_initialized()):
            return False
        return self.force or dist.get_world_size(self.group) > 1

    def on_ready(self, keys):
        """Backward hook: `keys` have just been enqueued on the current stream."""
        if not self.active() or self.defer:
            return
        for k in keys:
            i = self.key_bucket.get(k)
            if i is None:
                continue
            self.pending[i].discard(k)
            if not self.pending[i] and not self.launched[i]:
                a, b, _ = self.buckets[i]
                self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.launched[i] = True

    def finish(self):
        """Launch whatever is left and make the current stream wait for every bucket."""
        if self.active():
            for i, (a, b, _) in enumerate(self.buckets):
                if not self.launched[i]:
                    self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self.launched[i] = True
            for w in self.works:
                w.wait()
        self.reset()


def shard_indices(n_items, rank, world_size):
    """Disjoint stride-by-rank shard of a (shuffled) patch index list; every rank gets the same count
    (the tail is dropped) so all ranks run the same number of steps."""
    per = n_items // world_size
    return list(range(rank, per * world_size, world_size))


class ShardSampler(torch.utils.data.Sampler):
    """Epoch-aware disjoint shards of a dataset's indices (DistributedSampler's contract): every rank derives the SAME
    permutation from (seed, epoch) -- a private torch.Generator, independent of any per-process RNG state -- and takes
    positions rank, rank + world, ...  All ranks together visit every item at most once per epoch (the tail n % world is
    dropped so that every rank runs the same number of steps); call set_epoch(e) before each epoch for a new permutation.
    The reference's single-process DataParallel loop visits every patch each epoch (train.py:73-85)."""

    def __init__(self, n_items, rank=0, world_size=1, seed=0, shuffle=True):
        if not 0 <= rank < world_size:
            raise ValueError(f'rank {rank} outside world of {world_size}')
        self.n, self.rank, self.world, self.seed, self.shuffle = int(n_items), rank, world_size, int(seed), shuffle
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n // self.world

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed * 1000003 + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        return iter(order[i] for i in shard_indices(self.n, self.rank, self.world))


def allreduce_mean_grads(params, world_size, group=None, bucket_bytes=16 << 20):
    """Average the `.grad` of `params` over the ranks with a few large all-reduces instead of one blocking call per tensor
    (BiDateNet has 74 parameter tensors, 53.6 MB): gradients are packed into contiguous buckets of ~bucket_bytes in REVERSE
    parameter order (the order backward produced them), each bucket is reduced asynchronously while the next one is being
    packed, then scaled by 1 / world and copied back.  Used by the autograd training loop (train_epoch_autograd); the fused
    TrainStep reduces slices of its flat gradient buffer in place (GradBucketer) and needs no packing."""
    if world_size <= 1:
        return 0
    grads = [p.grad for p in reversed(list(params)) if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    inflight = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        inflight.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    inv = 1.0 / world_size
    for work, flat, b in inflight:
        work.wait()
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g)).mul_(inv)
            off += n
    return len(buckets)
