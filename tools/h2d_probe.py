"""Where does a host batch's way to the GPU spend its time on this box?  One 64-clip batch (N,3,300,25,2) fp32 = 11.5 MB:
pageable -> device (`.to(dev)`, what processor/recognition.py:258 does), pageable -> pinned (host copy), pinned -> device
(`non_blocking`), and harness.DeviceStager around a GPU workload of ~12 ms per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import istgcn_amd  # noqa
from istgcn_amd import harness
d = torch.device('cuda:0')
x = torch.randn(64, 3, 300, 25, 2)
y = torch.randint(0, 60, (64,))
mb = x.numel() * 4 / 1e6


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('batch %.1f MB, host threads %d' % (mb, torch.get_num_threads()))
ms = t(lambda: x.to(d)); print('pageable -> device (.to)         %7.2f ms  %6.1f GB/s' % (ms, mb / ms))
hp = torch.empty_like(x).pin_memory()
ms = t(lambda: hp.copy_(x)); print('pageable -> pinned (host copy)   %7.2f ms  %6.1f GB/s' % (ms, mb / ms))
dx = torch.empty_like(x, device=d)
ms = t(lambda: dx.copy_(hp, non_blocking=True)); print('pinned -> device (non_blocking)  %7.2f ms  %6.1f GB/s' % (ms, mb / ms))
torch.set_num_threads(1)
ms = t(lambda: hp.copy_(x)); print('pageable -> pinned, 1 host thread %6.2f ms  %6.1f GB/s' % (ms, mb / ms))
a = torch.randn(4096, 4096, device=d, dtype=torch.bfloat16)


def work(data):
    z = a
    for _ in range(110):                 # ~12 ms of queued matrix work per "step"
        z = z @ a
    return z.float().sum() + data.sum()


import types
stats = {'sync': 0.0, 'memmove': 0.0, 'enqueue': 0.0, 'n': 0}
_orig = harness.DeviceStager._stage


def timed_stage(self, i, data, label):
    hx, hy, dx, dy, ready, free = self._slot(i % self.depth, data, label)
    t0 = time.perf_counter(); free.synchronize(); t1 = time.perf_counter()
    harness._host_copy(hx, data); harness._host_copy(hy, label); t2 = time.perf_counter()
    with torch.cuda.stream(self.stream):
        dx.copy_(hx, non_blocking=True); dy.copy_(hy, non_blocking=True); ready.record(self.stream)
    t3 = time.perf_counter()
    stats['sync'] += t1 - t0; stats['memmove'] += t2 - t1; stats['enqueue'] += t3 - t2; stats['n'] += 1
    return dx, dy, ready, free


harness.DeviceStager._stage = timed_stage
yd = y.to(d)
for rep in range(2):
    for name, mk in (('resident batch', lambda: ((dx, yd) for _ in range(20))), ('DeviceStager', lambda: harness.DeviceStager(((x, y) for _ in range(20)), d))):
        for k in stats: stats[k] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0; tw = 0.0
        for data, label in mk():
            a0 = time.perf_counter(); s = work(data); tw += time.perf_counter() - a0; n += 1
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / n * 1e3
        extra = '' if not stats['n'] else '  | per step: free.synchronize %.2f ms, memmove %.2f, enqueue copies %.2f' % tuple(stats[k] / stats['n'] * 1e3 for k in ('sync', 'memmove', 'enqueue'))
        print('%-16s %6.2f ms per step over %d steps (host time issuing the work %.2f ms per step)%s' % (name, tot, n, tw / n * 1e3, extra))
