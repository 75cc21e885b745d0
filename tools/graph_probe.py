"""Experiment: how much of the step's dispatch overhead does hipGraph capture remove?  (Dropout seeds are frozen into the
captured launches here -- measurement only, not a training mode.)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import istgcn_amd
from istgcn_amd import harness
from istgcn_amd.net import st_gcn_msgcn
d = torch.device('cuda:0')
m = st_gcn_msgcn.Model(3, 60, {'layout': 'ntu-rgb+d', 'strategy': 'spatial_3'}, True, dropout=0.5, compute_dtype='bfloat16').to(d)
opt = harness.make_optimizer(m)
x = torch.randn(64, 3, 300, 25, 2, device=d); y = torch.randint(0, 60, (64,), device=d)
def step():
    return harness.train_step(m, opt, x, y)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager  %.2f ms/step' % timeit(step), flush=True)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
try:
    with torch.cuda.graph(g):
        loss = step()
    print('graph  %.2f ms/step' % timeit(g.replay), flush=True)
except Exception as e:
    print('capture failed:', repr(e)[:400])
