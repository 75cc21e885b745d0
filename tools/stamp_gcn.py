"""Diagnostic: per-phase cycle shares of the graph-conv kernels (needs a library built with -DISTGCN_STAMP)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import istgcn_amd
from istgcn_amd import ops, _lib
from istgcn_amd.net.utils.graph import Graph
lib = _lib.load()
d = torch.device('cuda:0'); dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
gr = Graph('ntu-rgb+d', 'spatial_3'); A = torch.tensor(gr.A + gr.A2 + gr.A3, dtype=torch.float32, device=d); cap = int((A != 0).sum())
names_b = ['tile setup', 'stage dy', 'barrier', 'GEMM', 'barrier', 'dxa->LDS + stage x', 'barrier', 'aggregate + dx store', 'dA dots', 'barrier']
for c, T in ((64, 300), (128, 150), (256, 75)):
    NM, V, K = 128, 25, 3
    x = torch.randn(NM, T, V, c, device=d).to(dt); dy = torch.randn(NM, T, V, c, device=d).to(dt)
    w3 = torch.randn(K, c, c, device=d) * 0.05
    f = lambda: ops.gcn_bwd_data(dy, A, w3, x=x, want_dA=True, nnz_cap=cap)
    f(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.istgcn_debug_stamps_gcn_bwd(out, 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    lib.istgcn_debug_stamps_gcn_bwd(out, 1)
    v = list(out); tot = sum(v[:10]) or 1
    print('C=%d gcn_bwd: ' % c + ', '.join('%s %.1f%%' % (n, 100 * x_ / tot) for n, x_ in zip(names_b, v[:10])), flush=True)
