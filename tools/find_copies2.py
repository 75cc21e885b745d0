"""Which Python lines of a training step issue device copies / fills (torch.profiler, stacks)?"""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import istgcn_amd  # noqa
from istgcn_amd import harness
from torch.profiler import profile, ProfilerActivity
from collections import Counter
dev = torch.device('cuda:0')
gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
m = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(3, 60, gargs, True, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
opt = harness.make_optimizer(m)
x = torch.randn(8, 3, 64, 25, 2, device=dev)
y = torch.randint(0, 60, (8,), device=dev)
for _ in range(3):
    harness.train_step(m, opt, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    harness.train_step(m, opt, x, y)
    torch.cuda.synchronize()
ev = prof.events()
dev_names = Counter(e.name[:60] for e in ev if e.device_type is not None and str(e.device_type).endswith('CUDA'))
print('device events:')
for k, v in dev_names.most_common(40):
    if any(t in k for t in ('opy', 'ill', 'emcpy', 'emset', 'elementwise')):
        print('  %4d %s' % (v, k))
c = Counter()
for e in ev:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::fill_', 'aten::zero_', 'aten::zeros'):
        st = [s.split('/')[-1] for s in (e.stack or []) if 'istgcn' in s or 'ist-gcn' in s or 'harness' in s][:3]
        par = e.cpu_parent.name if e.cpu_parent is not None else None
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else None
        c[(e.name, str(e.input_shapes)[:80], par, gp, tuple(st))] += 1
for k, v in c.most_common(40):
    print(v, k)
