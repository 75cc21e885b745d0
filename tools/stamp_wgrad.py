"""Diagnostic: per-phase cycle shares of the wgrad kernel (needs a library built with -DISTGCN_STAMP)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import istgcn_amd
from istgcn_amd import ops, _lib
from istgcn_amd.net.utils.graph import Graph
lib = _lib.load()
d = torch.device('cuda:0'); dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
gr = Graph('ntu-rgb+d', 'spatial_3'); A = torch.tensor(gr.A + gr.A2 + gr.A3, dtype=torch.float32, device=d); cap = int((A != 0).sum())
names = ['tile setup', 'prefetch issue / agg', 'dbias', 'MFMA loop', 'barrier', 'tail', 'flush', 'load wait', 's8', 's9', 's10', 's11', 's12', 's13']
# s8..s13 -- tconv: commit dz, urow, commit u, barrier | gcn: commit x, barrier, aggregate, barrier, commit dz+barrier, prefetch issue (s1 = S sums)
for c, T in ((64, 300), (128, 150), (256, 75)):
    NM, V, k = 128, 25, 9
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    pre = torch.stack([torch.ones(c), torch.zeros(c)]).to(d)
    for tag, f in (('tconv_wgrad', lambda: ops.tconv_wgrad(g, g, taps, in_mul=1, pre=pre, pre_relu=True)),
                   ('gcn_wgrad', lambda: ops.gcn_wgrad(g, g, A, nnz_cap=cap))):
        f(); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 16)()
        lib.istgcn_debug_stamps_wgrad(out, 1)
        for _ in range(5): f()
        torch.cuda.synchronize()
        lib.istgcn_debug_stamps_wgrad(out, 1)
        v = list(out); tot = sum(v[:14]) or 1
        print('C=%d %s: ' % (c, tag) + ', '.join('%s %.1f%%' % (n, 100 * x / tot) for n, x in zip(names, v[:14])), flush=True)
