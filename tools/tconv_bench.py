"""Times the temporal-conv forward / data-gradient (/ weight-gradient) launches of the trunk shapes (bench.py config 2:
NM = 128; 64 ch T=300, 128 ch T=150, 256 ch T=75) through the C ABI and prints one line: us per launch (best of 3
rounds of 10) and the fraction of the dense 16-bit MFMA peak.  A/B: run it once per setting of a dispatch override
(ISTGCN_TCONV_LEAN=0/1, ISTGCN_LIB_PATH=...), each in its own process -- the overrides are read once.
usage: tconv_bench.py [bf16|f16] [taps] [wgrad]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
with_wgrad = len(sys.argv) > 3 and sys.argv[3] == 'wgrad'
d = torch.device('cuda:0')
NM, V = 128, 25
out = []
gen = torch.Generator(device=d).manual_seed(0)
for c, T in ((64, 300), (128, 150), (256, 75)):
    P = NM * T * V
    g = torch.randn(NM, T, V, c, device=d, generator=gen).to(dt)
    dz = torch.randn(NM, T, V, c, device=d, generator=gen).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    wpt = ops.pack_tconv_weight(torch.randn(k, c, c, device=d, generator=gen) * (c * k) ** -0.5, V, taps, im, dt)
    pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
    bias = torch.zeros(c, device=d)
    st = ops.new_stats(c, d)
    maux = torch.cat([pre, torch.zeros(1, c, device=d), torch.ones(1, c, device=d)]).contiguous()
    fns = {'fwd': lambda: ops.tconv(g, wpt, c, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1),
           'bwd': lambda: ops.tconv(dz, wpt, c, taps, aux=g, maux=maux, stats=st, mode=1, Tout=T, Mlog=T, in_mul=1)}
    if with_wgrad:
        fns['wgrad'] = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=1, pre=pre, pre_relu=True)
    for name, fn in fns.items():
        best = 1e9
        for rnd in range(3):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        out.append('%dch %s %.0f us (%.2f)' % (c, name, best, 2.0 * P * c * c * k / best / 1e6 / 2500.0))
tag = ' '.join('%s=%s' % (e, os.environ[e]) for e in ('ISTGCN_TCONV_LEAN', 'ISTGCN_LIB_PATH') if e in os.environ) or 'default'
print('%-22s k=%d %s | ' % (tag, k, sys.argv[1] if len(sys.argv) > 1 else 'bf16') + ' | '.join(out), flush=True)
