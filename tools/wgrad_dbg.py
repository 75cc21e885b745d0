"""Per-phase cycle sums of workgroup (0,0) of the wave-specialised tconv_wgrad kernel (ISTGCN_WGRAD_DBG hook)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
for c, T in ((64, 300), (128, 150), (256, 75)):
    dz = (torch.randn(NM, T, V, c, device=d) * 0.1).to(dt)
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    pre = torch.stack([torch.ones(c), torch.zeros(c)]).to(d)
    fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    os.environ['ISTGCN_WGRAD_DBG'] = '1'
    fn()
    torch.cuda.synchronize()
    del os.environ['ISTGCN_WGRAD_DBG']
