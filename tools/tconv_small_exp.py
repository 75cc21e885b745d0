"""A/B of the two temporal-conv kernels on the bottleneck shapes of config 5 (st_gcn_mstcn_1x1_deep, NM=256, T=600, fp16):
run once with ISTGCN_TCONV_V1=0 and once with =1 (the choice is read once per process, before packing)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = torch.float16
d = torch.device('cuda:0')
NM, V = 256, 25
print('ISTGCN_TCONV_V1 =', os.environ.get('ISTGCN_TCONV_V1', '(rule)'))
for cin, cout, k, T in ((64, 8, 1, 600), (8, 8, 15, 600), (8, 64, 1, 600), (128, 11, 1, 300), (11, 11, 15, 300), (11, 128, 1, 300),
                        (256, 16, 1, 150), (16, 16, 15, 150), (16, 256, 1, 150), (32, 64, 9, 300), (64, 32, 9, 300)):
    g = torch.randn(NM, T, V, cin, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    wp = ops.pack_tconv_weight(torch.randn(k, cout, cin, device=d) * (cin * k) ** -0.5, V, taps, im, dt)
    bias = torch.zeros(cout, device=d)
    pre = torch.stack([torch.ones(cin), torch.zeros(cin)]).to(d)
    st = ops.new_stats(cout, d)
    fn = lambda: ops.tconv(g, wp, cout, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    nbytes = NM * T * V * (cin + cout) * 2
    print('%3d->%3d k=%2d T=%3d: %6.0f us  %5.2f TB/s' % (cin, cout, k, T, best, nbytes / best / 1e6), flush=True)
