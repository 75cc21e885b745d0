"""Run ONE library call a few times (for rocprofv3 counter passes over a single kernel).
usage: one_kernel.py twg <cin> <T> <stride> <taps> | gwg <cin> <cout> <T> | gbwd <cin> <cout> <T> | gfwd <cin> <cout> <T>"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
d = torch.device('cuda:0')
dt = torch.bfloat16
NM, V = 128, 25
what = sys.argv[1]
a = [int(v) for v in sys.argv[2:]]
if what == 'twg':
    cin, T, s, k = a
    Tz = (T + s - 1) // s
    dz = (torch.randn(NM, Tz, V, cin, device=d) * 0.1).to(dt)
    g = torch.randn(NM, T, V, cin, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, s)
    pre = torch.stack([torch.ones(cin), torch.zeros(cin)]).to(d)
    fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
else:
    cin, cout, T = a
    g_ = Graph('ntu-rgb+d', 'spatial_3')
    A = torch.tensor(g_.A + g_.A2 + g_.A3, dtype=torch.float32, device=d)
    cap = int((A != 0).sum())
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    dy = torch.randn(NM, T, V, cout, device=d).to(dt)
    W3 = (torch.randn(3 * cout, cin, device=d) * cin ** -0.5).view(3, cout, cin)
    wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)
    st = ops.new_stats(cout, d)
    fn = {'gfwd': lambda: ops.gcn_forward(x, A, wp, cout, stats=st, nnz_cap=cap),
          'gbwd': lambda: ops.gcn_bwd_data(dy, A, W3, x=x, addend=x if cin == cout else None, nnz_cap=cap),
          'gwg': lambda: ops.gcn_wgrad(dy, x, A, nnz_cap=cap)}[what]
for _ in range(5):
    fn()
torch.cuda.synchronize()
