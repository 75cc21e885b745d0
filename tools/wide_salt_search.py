"""For every (block kind, wide case) find the smallest input salt whose fixture has no ReLU pre-activation within `floor` of
zero (tests/golden/detinit.py: WIDE_X_SALT; tests/test_oracle_golden.py::test_wide_block_fixtures_have_no_knife_edge).
Runs the ORACLE (the fixture itself is then generated from the reference with that salt by make_golden.py g3w).
usage: python tools/wide_salt_search.py [floor=5e-6] [cases=3,4]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import detinit
from detinit import det_fill_, det_tensor, WIDE_BLOCKS, WIDE_T, wide_block_has
from oracle import stgcn_ref as R
from oracle.graph_ref import GraphRef
from test_oracle_golden import _block_adj

floor = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-6
cases = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(range(len(WIDE_BLOCKS)))
torch.set_num_threads(8)
for kind in ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn']:
    for si in cases:
        cin, cout, stride, V = WIDE_BLOCKS[si]
        if not wide_block_has(kind, si):
            continue
        g = GraphRef('ntu-rgb+d' if V == 25 else 'openpose', 'spatial_3')
        A, A2, A3 = (torch.tensor(a, dtype=torch.float32) for a in (g.A, g.A2, g.A3))
        K = A.shape[0]
        imps = [(0.5 + torch.rand((K, V, V), generator=torch.Generator().manual_seed(17 + j))) for j in range(3)]
        mst = (0.5 + torch.rand(3, generator=torch.Generator().manual_seed(21)))
        for salt in range(40):
            x = det_tensor('g3w.xw%d.' % si, (2, cin, WIDE_T, V), salt=salt)
            blk = R.RefBlock(kind, cin, cout, K, stride, dropout=0, residual=True)
            blk.load_state_dict(det_fill_(blk.state_dict(), salt=100 + si))
            blk.train()
            rec = {}
            first = blk.tcn[0] if blk.tcn_kind == 'single' else blk.tcn_start[0]
            last = blk.tcn[4] if blk.tcn_kind == 'single' else blk.tcn_end[1]
            h1 = first.register_forward_hook(lambda m, i, o: rec.__setitem__('bn1', o.detach()))
            h2 = last.register_forward_hook(lambda m, i, o: rec.__setitem__('y', o.detach()))
            with torch.no_grad():
                res = 0 if blk.res_mode == 'none' else (x if blk.res_mode == 'id' else blk.residual(x))
                blk(x, _block_adj(kind, A, A2, A3, imps), mst)
            h1.remove(); h2.remove()
            m1, m2 = float(rec['bn1'].abs().min()), float((rec['y'] + res).abs().min())
            if min(m1, m2) > floor:
                print("('%s', %d): %d,   # min |pre-activation| %.2e / %.2e" % (kind, si, salt, m1, m2), flush=True)
                break
        else:
            print(kind, si, 'no salt found')
