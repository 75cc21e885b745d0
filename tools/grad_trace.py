"""Debug aid: where does the backward signal of the HIP Model first deviate from the oracle?  Captures, per block,
dout (grad of the block output), dz (grad of the temporal-conv output) and dg (grad of the GCN output)."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import importlib
import istgcn_amd  # noqa
from istgcn_amd import ops
from oracle import stgcn_ref as R
from detinit import det_fill_, det_tensor, det_labels
tag = 'st_gcn_multi3_fix_3A_mstcn'
gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
dev = torch.device('cuda:0')
ref = R.RefModel(tag, 3, 60, gargs, True, dropout=0)
sd = det_fill_(ref.state_dict()); ref.load_state_dict(sd)
m = importlib.import_module('istgcn_amd.net.' + tag).Model(3, 60, gargs, True, dropout=0); m.load_state_dict(sd); m.to(dev)
x = det_tensor('smoke.x', (2, 3, 24, 25, 2)); y = det_labels('smoke.y', 2, 60)
ref.train(); m.train()
cap_r = {}
def hook_ref(blk, i):
    orig_gcn = blk.gcn.forward
    def gcn_fwd(x_, adj):
        g = orig_gcn(x_, adj); g.register_hook(lambda gr: cap_r.__setitem__(('dg', i), gr.permute(0, 2, 3, 1).clone())); return g
    blk.gcn.forward = gcn_fwd
    orig_end = blk.tcn_end.forward
    def end_fwd(z):
        z.register_hook(lambda gr: cap_r.__setitem__(('dz', i), gr.permute(0, 2, 3, 1).clone())); return orig_end(z)
    blk.tcn_end.forward = end_fwd
    orig = blk.forward
    def fwd(x_, adj, mst=None):
        o = orig(x_, adj, mst); o.register_hook(lambda gr: cap_r.__setitem__(('dout', i), gr.permute(0, 2, 3, 1).clone())); return o
    blk.forward = fwd
for i, b in enumerate(ref.st_gcn_networks):
    hook_ref(b, i)
F.cross_entropy(ref(x), y).backward()
cap_p = {}
order = {'n': len(m.st_gcn_networks)}
o_wg, o_twg = ops.gcn_wgrad, ops.tconv_wgrad
state = {'blk': len(m.st_gcn_networks)}
def p_twg(dz, g, taps, **kw):
    if len(taps) == 15:
        state['blk'] -= 1
        cap_p[('dz', state['blk'])] = dz.float().cpu().clone()
    return o_twg(dz, g, taps, **kw)
def p_wg(dy, x_, A, w3=None, **kw):
    cap_p[('dg', state['blk'])] = dy.float().cpu().clone(); return o_wg(dy, x_, A, w3, **kw)
ops.tconv_wgrad, ops.gcn_wgrad = p_twg, p_wg
for i, b in enumerate(m.st_gcn_networks):
    orig = b.run
    def run(x_, A, mst=None, nnz_cap=None, _o=orig, _i=i):
        o = _o(x_, A, mst, nnz_cap); o.register_hook(lambda gr: cap_p.__setitem__(('dout', _i), gr.float().cpu().clone())); return o
    b.run = run
F.cross_entropy(m(x.to(dev)), y.to(dev)).backward()
for i in reversed(range(10)):
    for k in ('dout', 'dz', 'dg'):
        a, b = cap_r[(k, i)].double(), cap_p[(k, i)].double()
        err = (a - b).abs()
        print('block %d %-4s max|.|=%9.3e relmax=%9.2e rell2=%9.2e  n(err>1e-3*max)=%d of %d' % (
            i, k, float(a.abs().max()), float(err.max() / a.abs().max()), float((a - b).norm() / a.norm()),
            int((err > 1e-3 * a.abs().max()).sum()), a.numel()))
