"""Per-phase cycle sums of workgroup 0 of the tconv kernel at the bench layer shapes (ISTGCN_TCONV_DBG hook)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
for c, T in ((64, 300), (128, 150), (256, 75)):
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
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        os.environ['ISTGCN_TCONV_DBG'] = '1'
        fn()
        torch.cuda.synchronize()
        del os.environ['ISTGCN_TCONV_DBG']
