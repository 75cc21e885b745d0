"""Times every launch of one bottleneck block (net/st_gcn_mstcn_1x1_deep.py:253-269) forward + backward at config-5 sizes
(NM=256, fp16): which of the narrow temporal convs / weight gradients carry the time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops, functional as Fn
dt = torch.float16
d = torch.device('cuda:0')
NM, V = 256, 25


def tm(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 200)
    return best


for C, w, T in ((64, 8, 600), (128, 11, 300), (256, 16, 150)):
    P = NM * T * V
    g = torch.randn(NM, T, V, C, device=d).to(dt)
    dz = (torch.randn(NM, T, V, C, device=d) * 0.1).to(dt)
    q = torch.randn(NM, T, V, w, device=d).to(dt)
    dq = (torch.randn(NM, T, V, w, device=d) * 0.1).to(dt)
    Ws = torch.randn(w, C, device=d) * C ** -0.5
    Wt = torch.randn(15, w, w, device=d) * (15 * w) ** -0.5
    We = torch.randn(C, w, device=d) * w ** -0.5
    pre = torch.stack([torch.ones(C), torch.zeros(C)]).to(d)
    coef = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(d)
    taps, im = ops.conv_taps_fwd(15, 1)
    ws = ops.pack_tconv_weight(Ws.view(1, w, C), V, [0], 1, dt)
    wt = ops.pack_tconv_weight(Wt, V, taps, im, dt)
    we = ops.pack_tconv_weight(We.view(1, C, w), V, [0], 1, dt)
    st = ops.new_stats(C, d)
    rows = [
        ('a  C->w 1 tap, BN+ReLU in', lambda: ops.tconv(g, ws, w, [0], pre=pre, pre_relu=True, Tout=T, Mlog=T), (C + w)),
        ('b  w->w 15 taps', lambda: ops.tconv(q, wt, w, taps, Tout=T, Mlog=T, in_mul=im), 2 * w),
        ('c  w->C 1 tap + BN sums', lambda: ops.tconv(q, we, C, [0], stats=st, Tout=T, Mlog=T), (C + w)),
        ('d  wgrad dz[C] x yb[w]', lambda: ops.tconv_wgrad(dz, q, [0], in_mul=1), (C + w)),
        ('e  data-grad C->w', lambda: Fn._conv_bwd_data(dz, We.view(1, C, w), 1, 1, T, w, V), (C + w)),
        ('f  wgrad 15 taps w x w', lambda: ops.tconv_wgrad(dq, q, taps, in_mul=im), 2 * w),
        ('g  data-grad 15 taps', lambda: Fn._conv_bwd_data(dq, Wt, 15, 1, T, w, V), 2 * w),
        ('h  wgrad dq[w] x g[C] (BN+ReLU in)', lambda: ops.tconv_wgrad(dq, g, [0], in_mul=1, pre=pre, pre_relu=True), (C + w)),
        ('i  data-grad w->C + mask + BN-bwd sums', lambda: Fn._conv_bwd_data(dq, Ws.view(1, w, C), 1, 1, T, C, V, aux=g, maux=coef, stats=st), (2 * C + w)),
    ]
    tot = 0.0
    for name, fn, ch in rows:
        us = tm(fn)
        tot += us
        print('C=%3d w=%2d T=%3d  %-42s %7.0f us  %5.2f TB/s' % (C, w, T, name, us, P * ch * 2 / us / 1e6), flush=True)
    print('C=%3d block total %.0f us' % (C, tot), flush=True)
