"""Per-call time (CUDA-graph replay) of tconv_wgrad at the trunk's layer shapes under ISTGCN_TWG_RC = 0 (round-2 kernels),
1 (frame-tiled kernel for stride-2 / 15-tap layers) and 2 (frame-tiled everywhere).  usage: twg_exp.py [bf16|f16] [taps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
d = torch.device('cuda:0')
NM, V = 128, 25
for cin, cout, T, s in ((64, 64, 300, 1), (128, 128, 150, 1), (256, 256, 75, 1), (128, 128, 300, 2), (256, 256, 150, 2)):
    Tz = (T + s - 1) // s
    dz = (torch.randn(NM, Tz, V, cout, device=d) * 0.1).to(dt)
    g = torch.randn(NM, T, V, cin, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, s)
    pre = torch.stack([torch.ones(cin), torch.zeros(cin)]).to(d)
    fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
    res = {}
    for mode in ('0', '2'):
        os.environ['ISTGCN_TWG_RC'] = mode
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(10):
                    fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record(st)
            gr.replay()
            e1.record(st)
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) * 100
    ref = ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
    os.environ['ISTGCN_TWG_RC'] = '0'
    old = ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
    err = float((ref[0] - old[0]).abs().max() / old[0].abs().max())
    errb = float((ref[1] - old[1]).abs().max() / old[1].abs().max())
    print('tconv_wgrad %d taps %3d->%3d T=%3d stride %d: round-2 %.1f us  frame-tiled %.1f us   (dW diff %.2e, dbias diff %.2e)' % (
        k, cin, cout, T, s, res['0'], res['2'], err, errb), flush=True)
