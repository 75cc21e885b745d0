"""Where does the HOST spend a training step of config 2 (no GPU sync inside the loop)?  Wall time per phase of
harness.train_step, plus a do-nothing phase right after it (a dict lookup + a tuple) that shows whatever stalls the thread
there (garbage collection, another thread holding the GIL).  usage: python tools/host_time_probe.py [gc_off]"""
import gc, importlib, os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import istgcn_amd  # noqa
from istgcn_amd import harness
mname, dts, B = bench.CONFIGS[2]
gargs, nc, V = bench.MODELS[mname]
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = importlib.import_module('istgcn_amd.net.' + mname).Model(3, nc, gargs, True, dropout=0.5, compute_dtype=torch.bfloat16)
model.apply(harness.weights_init)
model.to(dev).train()
opt = harness.make_optimizer(model)
x = torch.randn(B, 3, 300, V, 2).to(dev)
y = torch.randint(0, nc, (B,)).to(dev)
for _ in range(5):
    harness.train_step(model, opt, x, y)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == 'gc_off':
    gc.disable()
ph = {k: 0.0 for k in ('forward', 'loss', 'zero_grad', 'backward', 'step', 'nothing')}
d = {(i, (1, 2), (3,)): i for i in range(3)}
N = 30
t_all = time.perf_counter()
for it in range(N):
    t0 = time.perf_counter()
    out = model(x)
    t1 = time.perf_counter()
    loss = F.cross_entropy(out, y)
    t2 = time.perf_counter()
    opt.zero_grad()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    k = d.get((it % 3, tuple(x.shape[:2]), tuple(y.shape)))
    t6 = time.perf_counter()
    for name, a, b in (('forward', t0, t1), ('loss', t1, t2), ('zero_grad', t2, t3), ('backward', t3, t4), ('step', t4, t5), ('nothing', t5, t6)):
        ph[name] += b - a
torch.cuda.synchronize()
tot = (time.perf_counter() - t_all) / N * 1e3
print('gc %s: %.2f ms per step wall | host ms per step: %s' % ('off' if not gc.isenabled() else 'on', tot, ', '.join('%s %.2f' % (k, v / N * 1e3) for k, v in ph.items())))
print('gc counts', gc.get_count(), 'threads', torch.get_num_threads())
