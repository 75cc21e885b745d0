"""Per-phase cycle sums of workgroup (0,0) of the wave-specialised gcn_wgrad kernel (ISTGCN_WGRAD_DBG hook)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
g = Graph('ntu-rgb+d', 'spatial_3')
A = torch.tensor(g.A + g.A2 + g.A3, dtype=torch.float32, device=d)
cap = int((A != 0).sum())
NM, V = 128, 25
for cin, cout, T in ((64, 64, 300), (128, 128, 150), (256, 256, 75)):
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    dy = torch.randn(NM, T, V, cout, device=d).to(dt)
    fn = lambda: ops.gcn_wgrad(dy, x, A, nnz_cap=cap)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    os.environ['ISTGCN_WGRAD_DBG'] = '1'
    fn()
    torch.cuda.synchronize()
    del os.environ['ISTGCN_WGRAD_DBG']
