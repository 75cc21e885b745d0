"""Which Python lines launch the device-to-device copies of a training step (config 2, eager): torch.profiler with stacks."""
import os, sys, importlib, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import harness
import bench
dev = torch.device('cuda:0')
gargs, nc, V = bench.MODELS['st_gcn_msgcn']
model = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(3, nc, gargs, True, dropout=0.5, compute_dtype=torch.bfloat16)
model.apply(harness.weights_init)
model.to(dev).train()
opt = harness.make_optimizer(model, loss_scale=1.0)
x = torch.randn(16, 3, 300, V, 2).to(dev)
y = torch.randint(0, nc, (16,)).to(dev)
for _ in range(3):
    harness.train_step(model, opt, x, y, None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.train_step(model, opt, x, y, None)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::add_'):
        st = [s for s in (ev.stack or []) if 'istgcn_amd' in s or 'bench' in s or 'autograd' in s][:2]
        cnt[(ev.name, ' <- '.join(st) if st else '(no python frame: autograd engine)')] += 1
for (name, st), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print('%4d  %-18s %s' % (n, name, st))
