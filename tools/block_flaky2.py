"""The body of tests/test_gpu_block.py::test_blocks_wide_golden[st_gcn_msgcn-float32], case w0, with every output's error
against the reference fixture printed instead of asserted (which outputs are off when the test fails once in ten fresh
processes?), twice in the same process."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import istgcn_amd  # noqa
from istgcn_amd.net.utils.graph import Graph
from detinit import det_fill_, wide_block_inputs, subsample
kind = 'st_gcn_msgcn'
dt = torch.float32
d = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'block_g3w_%s.npz' % kind))
mod = importlib.import_module('istgcn_amd.net.' + kind)
gr = Graph('ntu-rgb+d', 'spatial_3')
A, A2, A3 = (torch.tensor(a, dtype=torch.float32, device=d) for a in (gr.A, gr.A2, gr.A3))
K = A.shape[0]
b = 'w0.'
x, r = wide_block_inputs(0, kind=kind)


def err(key, got):
    ref = torch.from_numpy(g[key]).double()
    got = got.detach().double().cpu()
    s = subsample(key, got, ref.numel()) if (key + '#norm') in g.files else got.reshape(ref.shape)
    return float((s - ref).abs().max() / max(1.0, float(ref.abs().max())))


# POISON=<value>: fill the allocator's pool with that float before anything runs -- a kernel that reads memory nobody
# wrote in this process then shows up every time instead of once in ten fresh processes
if os.environ.get('POISON'):
    junk = [torch.full((1 << 26,), float(os.environ['POISON']), device=d) for _ in range(6)]      # 1.5 GiB
    torch.cuda.synchronize()
    del junk
worst_all = 0.0
for rep in range(2):
    blk = mod.st_gcn(64, 64, (9, K), 1, dropout=0, residual=True)
    blk.load_state_dict(det_fill_(blk.state_dict(), salt=100), strict=True)
    blk.to(d)
    imps = [torch.from_numpy(g[b + 'imp%d' % j]).to(d).requires_grad_(True) for j in (1, 2, 3)]
    xin = x.to(d, dt)
    blk.eval()
    with torch.no_grad():
        y = blk(xin, A * imps[0], A2 * imps[1], A3 * imps[2])[0]
    res = {'y_eval': err(b + 'y_eval', y.float())}
    blk.train()
    xx = xin.clone().requires_grad_(True)
    y = blk(xx, A * imps[0], A2 * imps[1], A3 * imps[2])[0]
    res['y_train'] = err(b + 'y_train', y.float())
    (y.float() * r.to(d)).sum().backward()
    res['dx'] = err(b + 'dx', xx.grad.float())
    for k, p in blk.named_parameters():
        if b + 'grad.' + k in g.files and p.grad is not None:
            res['grad.' + k] = err(b + 'grad.' + k, p.grad)
    for j in (1, 2, 3):
        if b + 'dimp%d' % j in g.files:
            res['dimp%d' % j] = err(b + 'dimp%d' % j, imps[j - 1].grad)
    bad = {k: v for k, v in res.items() if v > 2e-4}
    worst_all = max(worst_all, max(res.values()))
    print('rep %d: %s' % (rep, 'all outputs within 2e-4 (worst %.1e)' % max(res.values()) if not bad else 'OFF: ' + ' '.join('%s %.1e' % kv for kv in bad.items())), flush=True)
