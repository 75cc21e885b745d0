"""Per-call time (CUDA-graph replay) and per-phase cycle sums of workgroup (0,0) of the wave-specialised tconv_wgrad kernel
(ISTGCN_WGRAD_DBG hook), for the trunk's unit-stride layers and its two stride-2 layers."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
for cin, cout, T, s in ((64, 64, 300, 1), (128, 128, 150, 1), (256, 256, 75, 1), (128, 128, 300, 2), (256, 256, 150, 2)):
    Tz = (T + s - 1) // s
    dz = (torch.randn(NM, Tz, V, cout, device=d) * 0.1).to(dt)
    g = torch.randn(NM, T, V, cin, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, s)
    pre = torch.stack([torch.ones(cin), torch.zeros(cin)]).to(d)
    fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
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
    print('tconv_wgrad %3d->%3d T=%3d stride %d: %.1f us' % (cin, cout, T, s, e0.elapsed_time(e1) * 100), flush=True)
    os.environ['ISTGCN_WGRAD_DBG'] = '1'
    fn()
    torch.cuda.synchronize()
    del os.environ['ISTGCN_WGRAD_DBG']
