"""Run-to-run stability of the hot-path kernels at bench size (NM = 128, V = 25): every op is launched `reps` times on the same
inputs; outputs without atomics in their path must be bit-identical, sums through atomics equal to 1e-5.  A kernel with a race
(round 4: the round-1 weight-gradient kernel) shows up here as a run that differs.
usage: stability_sweep.py [bf16|f16|f32] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, NM, V, K = torch.device('cuda:0'), 128, 25, 3
gr = Graph('ntu-rgb+d', 'spatial_3')
A = torch.tensor(gr.A + gr.A2 + gr.A3, dtype=torch.float32, device=d)
cap = int((A != 0).sum())
torch.manual_seed(0)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))
bad = 0


def check(name, fn, exact):
    global bad
    outs = []
    for _ in range(reps):
        o = fn()
        outs.append([t.clone() for t in (o if isinstance(o, (tuple, list)) else (o,)) if t is not None])
        torch.randn(1 << 22, device=d).sum().item()            # something else on the card in between
    worst = 0.0
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            worst = max(worst, 0.0 if torch.equal(a, b) else max(rel(a, b), 1e-30))
    ok = worst == 0.0 if exact else worst < 1e-5
    bad += not ok
    print('%-58s %s  worst run-to-run difference %.1e%s' % (name, 'ok ' if ok else 'BAD', worst, '' if exact else ' (atomics)'), flush=True)


for c, T in ((64, 300), (128, 150), (256, 75)):
    x = torch.randn(NM, T, V, c, device=d).to(dt)
    dy = (torch.randn(NM, T, V, c, device=d) * 0.1).to(dt)
    W3 = torch.randn(K, c, c, device=d) * c ** -0.5
    wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)
    pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
    bias = torch.zeros(c, device=d)
    check('gcn_fwd %dch' % c, lambda: ops.gcn_forward(x, A, wp, c, nnz_cap=cap), True)
    check('gcn_bwd_data %dch (dx; dA through atomics)' % c, lambda: ops.gcn_bwd_data(dy, A, W3, x=x, want_dA=True, nnz_cap=cap), False)
    check('gcn_wgrad %dch' % c, lambda: ops.gcn_wgrad(dy, x, A, nnz_cap=cap), False)
    for k in (9, 15):
        for s in (1, 2):
            taps, im = ops.conv_taps_fwd(k, s)
            Tz = (T + s - 1) // s
            wpt = ops.pack_tconv_weight(torch.randn(k, c, c, device=d) * (c * k) ** -0.5, V, taps, im, dt)
            dz = dy[:, ::s].contiguous()
            check('tconv fwd %dch %d taps stride %d' % (c, k, s),
                  lambda: ops.tconv(x, wpt, c, taps, bias=bias, pre=pre, pre_relu=True, Tout=Tz, Mlog=Tz, in_mul=im), True)
            check('tconv_wgrad %dch %d taps stride %d' % (c, k, s),
                  lambda: ops.tconv_wgrad(dz, x, taps, in_mul=im, pre=pre, pre_relu=True, want_bias=False), False)
            check('tconv_wgrad %dch %d taps stride %d + dbias' % (c, k, s),
                  lambda: ops.tconv_wgrad(dz, x, taps, in_mul=im, pre=pre, pre_relu=True), False)
        if True:
            taps, im = ops.conv_taps_fwd(k, 1)
            wpt = ops.pack_tconv_weight(torch.randn(k, c, c, device=d) * (c * k) ** -0.5, V, taps, im, dt)
            maux = torch.cat([pre, torch.zeros(1, c, device=d), torch.ones(1, c, device=d)]).contiguous()
            check('tconv data gradient %dch %d taps' % (c, k),
                  lambda: ops.tconv(dy, wpt, c, taps, aux=x, maux=maux, mode=1, Tout=T, Mlog=T, in_mul=1), True)
print('%d unstable' % bad)
sys.exit(1 if bad else 0)
