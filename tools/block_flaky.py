"""Is the FIRST backward pass of a block in a fresh process the same as the later ones?  One 64 -> 64 stride-1 V = 25 block of
net/st_gcn_msgcn in fp32 (the shape of tests/test_gpu_block.py::test_blocks_wide_golden[st_gcn_msgcn-dt0], which failed once in
ten fresh processes at the end of round 4): forward + backward four times on the same inputs, every gradient of run 1 against
run 2..4.  usage: block_flaky.py [kind] [f32|bf16]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import istgcn_amd  # noqa
from istgcn_amd.net.utils.graph import Graph
from detinit import det_fill_, wide_block_inputs
kind = sys.argv[1] if len(sys.argv) > 1 else 'st_gcn_msgcn'
dt = {'f32': torch.float32, 'bf16': torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else 'f32']
d = torch.device('cuda:0')
mod = importlib.import_module('istgcn_amd.net.' + kind)
gr = Graph('ntu-rgb+d', 'spatial_3')
A, A2, A3 = (torch.tensor(a, dtype=torch.float32, device=d) for a in (gr.A, gr.A2, gr.A3))
K = A.shape[0]
x, r = wide_block_inputs(0, kind=kind)
blk = mod.st_gcn(64, 64, (9, K), 1, dropout=0, residual=True)
blk.load_state_dict(det_fill_(blk.state_dict(), salt=100), strict=True)
blk.to(d).train()
imps = [(0.5 + torch.rand(K, 25, 25)).to(d).requires_grad_(True) for _ in range(3)]
xin = x.to(d, dt)
outs = []
for run in range(4):
    for p in list(blk.parameters()) + imps:
        p.grad = None
    xx = xin.clone().requires_grad_(True)
    args = (A * imps[0], A2 * imps[1], A3 * imps[2]) if kind == 'st_gcn_msgcn' else (A * imps[0],)
    y = blk(xx, *args)[0]
    (y.float() * r.to(d)).sum().backward()
    torch.cuda.synchronize()
    o = {'y': y.detach().float().clone(), 'dx': xx.grad.float().clone()}
    for k, p in blk.named_parameters():
        if p.grad is not None:
            o['grad.' + k] = p.grad.float().clone()
    for j, im in enumerate(imps):
        if im.grad is not None:
            o['dimp%d' % j] = im.grad.float().clone()
    outs.append(o)
rel = lambda a, b: float((a - b).abs().max() / max(1e-30, float(b.abs().max())))
bad = False
for k in outs[1]:
    e1 = rel(outs[0][k], outs[1][k]); e2 = rel(outs[2][k], outs[1][k]); e3 = rel(outs[3][k], outs[1][k])
    flag = max(e1, e2, e3) > (1e-4 if dt == torch.float32 else 2e-2)
    bad |= flag
    if flag:
        dd = (outs[0][k] - outs[1][k]).abs()
        idx = torch.nonzero(dd > 1e-3 * outs[1][k].abs().max())
        print('%-28s run1 vs run2 %.1e | run3 vs run2 %.1e | run4 vs run2 %.1e ; %d entries off in run 1, first %s last %s shape %s' % (
            k, e1, e2, e3, idx.shape[0], idx[0].tolist() if idx.shape[0] else None, idx[-1].tolist() if idx.shape[0] else None, tuple(dd.shape)))
print('UNSTABLE' if bad else 'stable')
