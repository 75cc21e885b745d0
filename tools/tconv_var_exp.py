"""Times the temporal-conv forward / data-gradient launches of the trunk shapes with the library selected by
ISTGCN_LIB_PATH (experiment builds of tools/build_variant.sh) under ISTGCN_TCONV_ABL in {0, 4 (no commit / store)}."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = torch.bfloat16
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
out = []
for c, T in ((64, 300), (128, 150), (256, 75)):
    P = NM * T * V
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    wpt = ops.pack_tconv_weight(torch.randn(k, c, c, device=d) * (c * k) ** -0.5, V, taps, im, dt)
    pre = torch.stack([torch.ones(c), torch.zeros(c)]).to(d)
    bias = torch.zeros(c, device=d)
    st = ops.new_stats(c, d)
    maux = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(d)
    fns = {'fwd': lambda: ops.tconv(g, wpt, c, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1),
           'bwd': lambda: ops.tconv(g, wpt, c, taps, aux=g, maux=maux, stats=st, mode=1, Tout=T, Mlog=T, in_mul=1)}
    for name, fn in fns.items():
        res = {}
        for rnd in range(3):
            for abl in ('0', '4'):
                os.environ['ISTGCN_TCONV_ABL'] = abl
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(abl, []).append(e0.elapsed_time(e1) * 100)
        out.append('%3dch %s %.0f/%.0f' % (c, name, min(res['0']), min(res['4'])))
print('%-10s ' % os.path.basename(os.environ.get('ISTGCN_LIB_PATH', 'default')) + ' | '.join(out), flush=True)
