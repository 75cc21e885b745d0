"""Per-shape time (CUDA-graph replay) and per-phase cycle sums of workgroup 0 of the wave-specialised gcn_bwd_data kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
g = Graph('ntu-rgb+d', 'spatial')
A = torch.tensor(g.A, dtype=torch.float32, device=d)
K = A.shape[0]
cap = int((A != 0).sum())
NM, V = 128, 25
for cin, cout, T in ((64, 64, 300), (64, 128, 150), (128, 128, 150), (128, 256, 75), (256, 256, 75)):
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    dy = torch.randn(NM, T, V, cout, device=d).to(dt)
    add = torch.randn(NM, T, V, cin, device=d).to(dt)
    W = torch.randn(K, cout, cin, device=d) * cin ** -0.5
    fn = lambda: ops.gcn_bwd_data(dy, A, W, x=x, addend=add, want_dA=True, nnz_cap=cap)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(10):
                fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        e0.record(st)
        gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    print('gcn_bwd_data %3d->%3d T=%3d: %.1f us' % (cin, cout, T, e0.elapsed_time(e1) * 100), flush=True)
    os.environ['ISTGCN_GCNBWD_DBG'] = '1'
    fn()
    torch.cuda.synchronize()
    del os.environ['ISTGCN_GCNBWD_DBG']
