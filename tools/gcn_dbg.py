"""Per-phase cycle sums of workgroup 0 of gcn_fwd at the bench layer shapes (kernel built with its ISTGCN_GCN_DBG hook)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
g = Graph('ntu-rgb+d', 'spatial_3')
A = torch.tensor(g.A + g.A2 + g.A3, dtype=torch.float32, device=d)
cap = int((A != 0).sum())
NM, V, K = 128, 25, 3
for cin, cout, T in ((64, 64, 300), (128, 128, 150), (256, 256, 75)):
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    W3 = (torch.randn(K * cout, cin, device=d) * cin ** -0.5).view(K, cout, cin)
    wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)
    bterm = torch.randn(V, cout, device=d)
    st = ops.new_stats(cout, d)
    for _ in range(3):
        ops.gcn_forward(x, A, wp, cout, bterm=bterm, stats=st, nnz_cap=cap)
    torch.cuda.synchronize()
    os.environ['ISTGCN_GCN_DBG'] = '1'
    ops.gcn_forward(x, A, wp, cout, bterm=bterm, stats=st, nnz_cap=cap)
    torch.cuda.synchronize()
    del os.environ['ISTGCN_GCN_DBG']
