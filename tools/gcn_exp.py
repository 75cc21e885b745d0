"""Times gcn_fwd / gcn_bwd_data / gcn_wgrad at the bench layer shapes under environment switches, interleaved in one process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
which = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
envname = sys.argv[3] if len(sys.argv) > 3 else 'ISTGCN_GCN_TR'
vals = sys.argv[4].split(',') if len(sys.argv) > 4 else ['128', '64']
d = torch.device('cuda:0')
g = Graph('ntu-rgb+d', 'spatial_3')
A = torch.tensor(g.A + g.A2 + g.A3, dtype=torch.float32, device=d)
cap = int((A != 0).sum())
NM, V, K = 128, 25, 3
for cin, cout, T in ((3, 64, 300) if dt != torch.float32 else (64, 64, 300), (64, 64, 300), (64, 128, 300), (128, 128, 150), (128, 256, 150), (256, 256, 75)):
    P = NM * T * V
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    dy = torch.randn(NM, T, V, cout, device=d).to(dt)
    W3 = (torch.randn(K * cout, cin, device=d) * cin ** -0.5).view(K, cout, cin)
    wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)
    bterm = torch.randn(V, cout, device=d)
    st = ops.new_stats(cout, d)
    fns = {'fwd': lambda: ops.gcn_forward(x, A, wp, cout, bterm=bterm, stats=st, nnz_cap=cap),
           'bwd': lambda: ops.gcn_bwd_data(dy, A, W3, x=x, addend=x if cin == cout else None, nnz_cap=cap, want_dx=cin != 3),
           'wgrad': lambda: ops.gcn_wgrad(dy, x, A, nnz_cap=cap)}
    fn = fns[which]
    res = {}
    graphs = {}
    for v in vals:                       # one CUDA graph of 10 calls per setting: pure GPU time, no host launch gaps
        os.environ[envname] = v
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(10):
                fn()
        graphs[v] = g_
    for rnd in range(3):
        for v in vals:
            graphs[v].replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graphs[v].replay()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 100)
    print('%3d->%3d T=%3d %s: ' % (cin, cout, T, which) + '  '.join('%s=%s %.0f us' % (envname[-6:], v, min(t)) for v, t in res.items()), flush=True)
