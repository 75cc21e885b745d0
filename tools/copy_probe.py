"""Which host-side ops of one training step launch the small device copies / fills that rocprofv3 shows next to the kernels
(`__amd_rocclr_copyBuffer`, FillFunctor, direct_copy)?  One profiled step under torch.profiler; prints (1) the device-side
activity names with counts, (2) for every aten::copy_ / aten::clone / aten::fill_ / aten::zero_ the chain of enclosing ops up
to the autograd node or Python frame that issued it, with tensor shapes.
usage: python tools/copy_probe.py [--config 2]"""
import argparse, collections, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import istgcn_amd  # noqa
from istgcn_amd import harness

ap = argparse.ArgumentParser()
ap.add_argument('--config', type=int, default=2)
ap.add_argument('--batch', type=int, default=None)
a = ap.parse_args()
mname, dts, B = bench.CONFIGS[a.config]
B = a.batch or B
gargs, nc, V = bench.MODELS[mname]
T = 600 if mname.endswith('deep') else 300
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dts]
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = importlib.import_module('istgcn_amd.net.' + mname).Model(3, nc, gargs, True, dropout=0.5, compute_dtype=dt)
model.apply(harness.weights_init)
model.to(dev).train()
opt = harness.make_optimizer(model, loss_scale=65536.0 if dt == torch.float16 else 1.0)
x = torch.randn(B, 3, T, V, 2).to(dev)
y = torch.randint(0, nc, (B,)).to(dev)
for _ in range(3):
    harness.train_step(model, opt, x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    harness.train_step(model, opt, x, y)
    torch.cuda.synchronize()
ev = list(prof.events())
dev_names = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU:
        dev_names[e.name[:70]] += 1
print('--- device activities (count, name), %d kinds' % len(dev_names))
for n, c in dev_names.most_common(25):
    print('%5d  %s' % (c, n))
chains = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone'):
        p = e.cpu_parent
        if p is not None and p.name in ('aten::clone', 'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::zero_', 'aten::zeros',
                                        'aten::zeros_like') and e.name != 'aten::clone':
            pass
        names = [e.name]
        while p is not None and len(names) < 6:
            names.append(p.name[:60])
            p = p.cpu_parent
        frame = next((s for s in (e.stack or []) if ROOT in s and 'copy_probe' not in s), '')
        chains[(' <- '.join(names), str(e.input_shapes)[:60], frame.replace(ROOT + '/', '')[:80])] += 1
print('--- host ops that copy / fill: count, chain (innermost first), shapes, repo frame')
for (chain, shp, fr), c in chains.most_common(70):
    print('%4d  %s | %s | %s' % (c, chain, shp, fr))
