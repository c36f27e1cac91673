"""Which HIP streams share a hardware queue?  (test infrastructure)  Streams are created in order -- n normal-priority and h high-priority ones through
bdn_stream_create, plus torch's default stream -- and every pair is probed: a long sleep kernel on A, a tiny kernel on B; if B's kernel finishes only
after A's, the two are serialised, i.e. they sit on one hardware queue.   python tools/archive/probe_queues.py [n_normal=6] [n_high=3]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import _lib, streams
n_norm = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_high = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
def mk(prio):
    h = C.c_void_p(); _lib.call('bdn_stream_create', prio, C.byref(h))
    return torch.cuda.ExternalStream(h.value, device=dev)
ss = [('default', torch.cuda.default_stream(dev))]
order = ['n'] * n_norm + ['h'] * n_high
for i, k in enumerate(order):
    ss.append((f'{k}{i}', mk(1 if k == 'h' else 0)))
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES', 'unset'))
names = [n for n, _ in ss]
print('     ' + ' '.join(f'{n:>7s}' for n in names))
for na, a in ss:
    row = []
    for nb, b in ss:
        row.append('   -   ' if a is b else ('  SAME ' if streams.serialised(a, b) else '   .   '))
    print(f'{na:>5s}' + ' '.join(row))
