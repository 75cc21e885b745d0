"""Per-call time (hipGraph replay of 10 calls) of tconv_wgrad at the trunk's stride-1 layer shapes (NM = 128, V = 25), with and
without the conv-bias gradient (both on the lean kernel, tconv_wgrad_lean.hip -- with: plus its column-sum kernel;
ISTGCN_TWG_LEAN=0 in the environment: the round-1/2/3 kernels for both; one process per setting).  usage: twg_bench.py [bf16|f16] [taps]"""
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
for c, T, s in ((64, 300, 1), (128, 150, 1), (256, 75, 1), (128, 300, 2), (256, 150, 2)):
    dz = (torch.randn(NM, (T + s - 1) // s, V, c, device=d) * 0.1).to(dt)
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, s)
    pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
    res = {}
    outs = {}
    for wb in (True, False):
        fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True, want_bias=wb)
        for _ in range(3):
            outs[wb] = fn()
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
        res[wb] = e0.elapsed_time(e1) * 100
    err = float((outs[True][0] - outs[False][0]).abs().max() / outs[True][0].abs().max())
    fl = 2.0 * NM * ((T + s - 1) // s) * V * c * c * k
    print('tconv_wgrad %d taps %3d ch T=%3d stride %d: with dbias %.1f us (%.0f TFLOP/s)  without %.1f us (%.0f TFLOP/s)   dW diff %.2e' % (
        k, c, T, s, res[True], fl / res[True] * 1e-6, res[False], fl / res[False] * 1e-6, err), flush=True)
