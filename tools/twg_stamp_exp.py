"""Cycle stamps of one compute wave and one memory wave of workgroup 0 in the lean tconv weight gradient (experiment build
-DISTGCN_TWG_STAMP of tconv_wgrad_lean.hip, selected with ISTGCN_LIB_PATH): where a launch's time goes, per role."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = torch.bfloat16
d = torch.device('cuda:0')
NM, V, k = 128, 25, int(sys.argv[1]) if len(sys.argv) > 1 else 9
dbg = torch.zeros(16, dtype=torch.int64, device=d)
os.environ['ISTGCN_TWG_DBG_PTR'] = str(dbg.data_ptr())
for c, T in ((64, 300), (128, 150), (256, 75)):
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    dz = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
    fn = lambda: ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True, want_bias=False)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    v = dbg.cpu().tolist()
    nt = max(1, v[7])
    print('%3dch  %.0f us, %d tiles per workgroup; cycles per tile: compute wave 0: steps %.0f, own LDS reads %.0f, barrier %.0f, next tile\'s first reads + last step %.0f | '
          'compute wave 3: %.0f, %.0f, %.0f, %.0f | memory wave: issue %.0f, commit %.0f, barrier %.0f' % (
              c, e0.elapsed_time(e1) * 1e3, v[7], v[0] / nt, v[2] / nt, v[3] / nt, v[1] / nt, v[11] / nt, v[13] / nt, v[14] / nt, v[12] / nt,
              v[9] / nt, v[8] / nt, v[10] / nt), flush=True)
    print('        s_memtime %d ticks over %d ticks of the 100 MHz s_memrealtime: %.0f MHz' % (v[4], v[5], v[4] / max(1, v[5]) * 100.0), flush=True)
