"""Debug aid: per-parameter gradient error of the HIP Model vs the oracle on one tiny train-mode step."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import importlib
import istgcn_amd  # noqa
from oracle import stgcn_ref as R
from detinit import det_fill_, det_tensor, det_labels
tag = sys.argv[1] if len(sys.argv) > 1 else 'st_gcn_multi3_fix_3A_mstcn'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
dev = torch.device('cuda:0')
ref = R.RefModel(tag, 3, 60, gargs, True, dropout=0)
sd = det_fill_(ref.state_dict()); ref.load_state_dict(sd)
m = importlib.import_module('istgcn_amd.net.' + tag).Model(3, 60, gargs, True, dropout=0); m.load_state_dict(sd); m.to(dev)
x = det_tensor('smoke.x', (2, 3, T, 25, 2)); y = det_labels('smoke.y', 2, 60)
ref.train(); m.train()
F.cross_entropy(ref(x), y).backward()
F.cross_entropy(m(x.to(dev)), y.to(dev)).backward()
for (k, p), (_, q) in zip(ref.named_parameters(), m.named_parameters()):
    if p.grad is None:
        continue
    a, b = p.grad.double(), q.grad.double().cpu()
    print('%-50s max|g|=%9.3e  relmax=%9.2e  rell2=%9.2e' % (k, float(a.abs().max()), float((a - b).abs().max() / max(1e-30, float(a.abs().max()))), float((a - b).norm() / max(1e-30, float(a.norm())))))
