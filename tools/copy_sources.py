"""Where do the device-to-device copies of a training step come from?  Counts `aten::copy_` / `aten::clone` calls of one step by
Python call site (torch.profiler, with_stack)."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import harness
from istgcn_amd.net import st_gcn_msgcn as prod
d = torch.device('cuda:0')
m = prod.Model(3, 60, dict(layout='ntu-rgb+d', strategy='spatial_3'), True, dropout=0.5, compute_dtype=torch.bfloat16).to(d).train()
opt = harness.make_optimizer(m)
x = torch.randn(8, 3, 64, 25, 2, device=d)
y = torch.randint(0, 60, (8,), device=d)
for _ in range(3):
    harness.train_step(m, opt, x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.train_step(m, opt, x, y)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::add_', 'aten::mul', 'aten::add', 'aten::div', 'aten::sum'):
        st = [s for s in (e.stack or []) if 'ist-gcn_amd' in s or 'istgcn' in s or 'autograd' in s]
        cnt[(e.name, st[0] if st else '?')] += 1
for (name, where), n in cnt.most_common(40):
    print('%4d  %-18s %s' % (n, name, where))
kc = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print('device-side events:', sum(kc.values()))
for k, n in kc.most_common(12):
    print('%4d  %s' % (n, k[:100]))
