"""The three HIP streams of the training step, ONE set per device for the whole process.

The reference runs everything on PyTorch's default stream (train.py:83-101).  Here a step uses
  'chain'  high priority: forward, loss, the dz chain of backward, SGD (fabric_amd/train_step.py);
  'wgrad'  the weight-gradient GEMMs and the gradient all-reduce buckets launched behind them (fabric_amd/engine.py);
  'copy'   host -> device input copies (fabric_amd/input_pipeline.py, fabric_amd/utils/inference.py); 'copy2' is a second one that only
           the scene upload uses (one DMA engine per date).
They must sit on three different hardware queues (two streams on one queue serialise: +8-14 % step time).  Round 2 took
`torch.cuda.Stream()` per object, i.e. the NEXT stream of torch's pool each time, and some pool streams share a queue: the
third TrainStep of a process was slower than the first.  Now the library creates each role's stream once
(bdn_stream_create -> hipStreamCreateWithPriority) and every TrainStep / engine / feeder of the process shares it, so the
N-th object runs exactly like the first (tests/test_gpu_train.py::test_three_train_steps_same_speed).  Two steps issued
concurrently from two host threads on the same device therefore serialise -- one process per GPU, one step at a time, is
the deployment.
"""
import ctypes as C
import threading

import torch

from . import _lib

_ROLES = {'chain': 1, 'wgrad': 0, 'copy': 0, 'copy2': 0}        # role -> bdn_stream_create priority
_streams = {}
_lock = threading.Lock()


def get(role, device=None):
    """The process-wide stream of `role` on `device` (default: the current device) as a torch stream object."""
    if role not in _ROLES:
        raise ValueError(f'unknown stream role {role!r}: one of {sorted(_ROLES)}')
    if not torch.cuda.is_available():
        raise RuntimeError('fabric_amd: streams need a ROCm device -- there is no CPU path')
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError('fabric_amd: streams need a ROCm device -- there is no CPU path')
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, role)
    with _lock:
        s = _streams.get(key)
        if s is None:
            s = _streams[key] = _create_distinct(idx, role)
        return s


_graveyard = []          # NEVER-USED streams that collided with another role's hardware queue while a role stream was being created:
                         # kept alive so that their queue slot stays taken
_GRAVEYARD_CAP = 16      # such streams per process; beyond that the oldest is destroyed (nothing ever ran on it but the probe)
_retired = []            # former ROLE streams (replace()): work, events and a caller's `torch.cuda.stream(...)` scope may still refer to
                         # them, so they are never destroyed before the process exits


def _park(s):
    _graveyard.append(s)
    while len(_graveyard) > _GRAVEYARD_CAP:
        _destroy(_graveyard.pop(0))


def _destroy(s):
    try:
        s.synchronize()
        _lib.load().bdn_stream_destroy(s.cuda_stream)
    except Exception:
        pass


def _bury_all():
    """atexit: parked streams are destroyed (the role streams in use stay with the process until the HIP runtime tears down)."""
    while _graveyard:
        _destroy(_graveyard.pop())
    while _retired:
        _destroy(_retired.pop())


import atexit
atexit.register(_bury_all)


def _create(idx, role):
    with torch.cuda.device(idx):
        torch.cuda.current_stream()                    # the device's context exists before the library asks HIP for a stream
        h = C.c_void_p()
        _lib.call('bdn_stream_create', _ROLES[role], C.byref(h))
    s = torch.cuda.ExternalStream(h.value, device=torch.device('cuda', idx))
    with torch.cuda.stream(s):                         # first use creates the hardware queue (milliseconds): do it now, not inside a probe
        torch.zeros(1, device=s.device).add_(1.0)
    s.synchronize()
    return s


def _create_distinct(idx, role, tries=6):
    """A new stream for `role` that shares its hardware queue with none of the device's other role streams.  HIP multiplexes streams
    onto GPU_MAX_HW_QUEUES (default 4) hardware queues per priority class in creation order (tools/archive/probe_queues.py: the 5th and 6th
    normal-priority streams of a process land on the queues of the 3rd and 4th), so a stream created late may be serialised with the
    chain or the weight-gradient stream.  Checked with `serialised` against the library's OTHER role streams of the device and torch's
    default stream; a colliding stream is parked (it keeps its slot) and another one is created.  Streams the library cannot see -- a
    data loader's, RCCL's collective stream -- are not probed here: TrainStep.guard_collectives measures their effect on the step
    instead and calls replace() when it finds one."""
    others = [v for (i, r), v in _streams.items() if i == idx and r != role] + [torch.cuda.default_stream(idx)]
    s = _create(idx, role)
    for _ in range(tries):
        if not any(serialised(o, s) or serialised(s, o) for o in others):
            return s
        _park(s)
        s = _create(idx, role)
    import warnings
    warnings.warn(f'fabric_amd: no hardware queue of its own found for the {role!r} stream after {tries} tries: it shares one with another '
                  f'stream of the training step, which will serialise them (GPU_MAX_HW_QUEUES={__import__("os").environ.get("GPU_MAX_HW_QUEUES", "4 (default)")})',
                  RuntimeWarning)
    return s


def replace(role, device=None):
    """Retire the current stream of `role` and create a new one (TrainStep.guard_collectives: the role's stream turned out to be slowed
    down by a stream the library does not own, e.g. RCCL's collective stream).  Nothing in the library caches a role stream across
    calls (engines, TrainStep.stream() and the feeders fetch it on every use); work already queued on the retired stream completes, and
    the stream object stays valid for whoever still holds it (it is only destroyed at process exit)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        old = _streams.pop((idx, role), None)
        if old is not None:
            _retired.append(old)                      # stays alive: the caller may still be inside `with torch.cuda.stream(old)`
        s = _streams[(idx, role)] = _create_distinct(idx, role)
        return s


def restore(role, stream, device=None):
    """Make `stream` -- a former stream of `role`, as returned by get() before a replace() -- the role's stream again (the guard
    reverts to the arrangement that measured best).  The stream it displaces is retired, never destroyed."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        cur = _streams.get((idx, role))
        if cur is not None and cur.cuda_stream == stream.cuda_stream:
            return cur
        if cur is not None:
            _retired.append(cur)
        if stream in _retired:
            _retired.remove(stream)
        _streams[(idx, role)] = stream
        return stream


def serialised(a, b, sleep_cycles=6_000_000):
    """Do streams `a` and `b` share a hardware queue?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
    default); two streams on one queue run their kernels one after the other, which for the streams of the training step means
    +8-60 % step time.  Probe: a ~3 ms sleep kernel on `a`, a tiny kernel on `b`; if `b`'s kernel completes only when `a`'s has,
    they are serialised.  Synchronises the device; meant for setup time."""
    dev = a.device
    torch.cuda.synchronize(dev)
    t = torch.zeros(64, device=dev)
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(sleep_cycles)
        ea.record(a)
    with torch.cuda.stream(b):
        t.add_(1.0)
        eb.record(b)
    eb.synchronize()
    same = ea.query()                       # the sleep is already over when b's tiny kernel has finished: b waited behind it
    torch.cuda.synchronize(dev)
    return bool(same)


class HandOff:
    """A reusable device-local event (bdn_event_create: no timing, no system-scope fence): `signal(src)` then `wait(dst)` orders
    everything `dst` enqueues afterwards behind what `src` had enqueued.  Re-recording is safe: a wait captures the record that
    preceded it."""

    def __init__(self):
        h = C.c_void_p()
        _lib.call('bdn_event_create', C.byref(h))
        self._h = h.value

    def signal(self, src):
        _lib.call('bdn_event_record', self._h, src.cuda_stream)

    def wait(self, dst):
        _lib.call('bdn_stream_wait_event', dst.cuda_stream, self._h)

    def __del__(self):
        try:
            if self._h:
                _lib.load().bdn_event_destroy(self._h)
        except Exception:
            pass
