"""Per-launch times of the register-chained bottleneck kernels (csrc/bneck_rc.hip) at config-5 sizes (NM=256, fp16) next to the
narrow 15-tap weight gradient that still runs the generic kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
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
    wp = (w + 7) // 8 * 8
    P = NM * T * V
    g = torch.randn(NM, T, V, C, device=d).to(dt)
    dz = (torch.randn(NM, T, V, C, device=d) * 0.1).to(dt)
    q = torch.randn(NM, T, V, wp, device=d).to(dt)
    Ws = torch.randn(w, C, device=d) * C ** -0.5
    Wt = torch.randn(15, w, w, device=d) * (15 * w) ** -0.5
    We = torch.randn(C, w, device=d) * w ** -0.5
    pre = torch.stack([torch.ones(C), torch.zeros(C)]).to(d)
    coef = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(d)
    taps, im = ops.conv_taps_fwd(15, 1)
    st = ops.new_stats(C, d)
    yb = torch.empty(NM, T, V, wp, device=d, dtype=dt)
    z = torch.empty(NM, T, V, C, device=d, dtype=dt)
    tl = sorted(ops.conv_taps_bwd(15, 1, 0), key=lambda jd: jd[1])
    rows = [
        ('bneck_in  C->w, BN+ReLU in', lambda: ops.bneck_in(g, Ws, wp, pre=pre, pre_relu=True), (C + wp)),
        ('bneck_in  C->w (dyb = We^T dz)', lambda: ops.bneck_in(dz, We.t(), wp), (C + wp)),
        ('bneck_out w->15 taps->w->C + sums', lambda: ops.bneck_out(q, Wt, list(range(15)), -7, We, C, stats=st, mode=0, Tout=T, Mlog=T, yb=yb, z=z), (2 * wp + C)),
        ('bneck_out mode 1 (mask + bwd sums)', lambda: ops.bneck_out(q, Wt.transpose(1, 2), [j for j, _ in tl], tl[0][1], Ws.t(), C, aux=g, maux=coef, stats=st, mode=1, Tout=T, Mlog=T, yb=yb, z=z), (2 * wp + 2 * C)),
        ('bneck_wgrad dz[C] x yb[w]', lambda: ops.bneck_wgrad(dz, q, True), (C + wp)),
        ('bneck_wgrad dq[w] x g[C] (BN+ReLU in)', lambda: ops.bneck_wgrad(g, q, False, pre=pre, pre_relu=True), (C + wp)),
        ('generic wgrad 15 taps w x w', lambda: ops.tconv_wgrad(q, q, taps, in_mul=im), 2 * wp),
    ]
    tot = 0.0
    for name, fn, ch in rows:
        us = tm(fn)
        tot += us
        print('C=%3d w=%2d T=%3d  %-42s %7.0f us  %5.2f TB/s' % (C, w, T, name, us, P * ch * 2 / us / 1e6), flush=True)
    print('C=%3d block total %.0f us' % (C, tot), flush=True)
