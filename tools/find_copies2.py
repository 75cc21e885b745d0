import sys, os, torch, importlib
sys.path.insert(0, '/root/repo')
import istgcn_amd
from istgcn_amd import harness
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
m = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(3, 60, gargs, True, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
opt = harness.make_optimizer(m)
x = torch.randn(8, 3, 64, 25, 2, device=dev); y = torch.randint(0, 60, (8,), device=dev)
for _ in range(3): harness.train_step(m, opt, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.train_step(m, opt, x, y)
    torch.cuda.synchronize()
ev = prof.events()
n = 0
for e in ev:
    if 'emcpy' in e.name or 'copyBuffer' in e.name:
        n += 1
# group CPU ops that are aten::copy_ / aten::contiguous / clone with their stack top
from collections import Counter
c = Counter()
for e in ev:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::to'):
        st = [s for s in (e.stack or []) if 'repo' in s or 'autograd' in s][:2]
        c[(e.name, tuple(st))] += 1
print('memcpy-like device events:', n)
for k, v in c.most_common(25):
    print(v, k)
