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


for name, it in (('resident batch', ((dx, y.to(d)) for _ in range(20))), ('DeviceStager', harness.DeviceStager(((x, y) for _ in range(20)), d))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for data, label in it:
        s = work(data); n += 1
    torch.cuda.synchronize()
    print('%-16s %6.2f ms per step over %d steps' % (name, (time.perf_counter() - t0) / n * 1e3, n))
