"""Experiment driver for the temporal-conv kernel: times forward / data-gradient launches at the bench layer shapes under
diagnostic environment switches (ISTGCN_TCONV_WM: compute-wave layout, ISTGCN_TCONV_ABL: 1 = no input loads, 2 = no MFMAs),
interleaved in one process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops

dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
VARIANTS = [('full', {'ISTGCN_TCONV_ABL': '0'}), ('no-loads', {'ISTGCN_TCONV_ABL': '1'}), ('no-mfma', {'ISTGCN_TCONV_ABL': '2'}),
            ('w@L1', {'ISTGCN_TCONV_ABL': '3'}), ('no-commit/store', {'ISTGCN_TCONV_ABL': '4'})]
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
            for tag, env in VARIANTS:
                os.environ.update(env)
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(tag, []).append(e0.elapsed_time(e1) * 100)
        fl = 2.0 * P * c * c * k
        print('%3dch %s: ' % (c, name) + '  '.join('%s %.0f us (%.0f TF)' % (t, min(v), fl / min(v) / 1e6) for t, v in res.items()), flush=True)
os.environ.update({'ISTGCN_TCONV_ABL': '0'})
