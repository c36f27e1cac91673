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
            with torch.cuda.device(idx):
                torch.cuda.current_stream()            # the device's context exists before the library asks HIP for a stream
                h = C.c_void_p()
                _lib.call('bdn_stream_create', _ROLES[role], C.byref(h))
            s = _streams[key] = torch.cuda.ExternalStream(h.value, device=torch.device('cuda', idx))
        return s


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
