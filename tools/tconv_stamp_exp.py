"""Cycle stamps of one compute wave and one memory wave of workgroup 0 in `tconv` (experiment build -DISTGCN_TCONV_STAMP,
selected with ISTGCN_LIB_PATH): where a launch's time goes, per role, for the trunk shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
dt = torch.bfloat16
d = torch.device('cuda:0')
NM, V, k = 128, 25, 9
dbg = torch.zeros(16, dtype=torch.int64, device=d)
os.environ['ISTGCN_TCONV_DBG_PTR'] = str(dbg.data_ptr())
CN = ['tile start + ring prologue', 'steps', 'wait item barrier', 'acc -> image', 'wait image barrier', 'prologue']
MN = ['issue (+aux)', 'commit', 'epilogue part', 'wait item + image barrier', 'wait for the chunk loads', 'prologue']
for c, T in ((64, 300), (128, 150), (256, 75)):
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    dz = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    wpt = ops.pack_tconv_weight(torch.randn(k, c, c, device=d) * (c * k) ** -0.5, V, taps, im, dt)
    pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
    bias = torch.zeros(c, device=d)
    st = ops.new_stats(c, d)
    maux = torch.cat([pre, torch.zeros(1, c, device=d), torch.ones(1, c, device=d)]).contiguous()
    fns = {'fwd': lambda: ops.tconv(g, wpt, c, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1),
           'bwd': lambda: ops.tconv(dz, wpt, c, taps, aux=g, maux=maux, stats=st, mode=1, Tout=T, Mlog=T, in_mul=1)}
    for name, fn in fns.items():
        for abl in ('0',):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            v = dbg.cpu().tolist()
            ct, mt = sum(v[0:6]), sum(v[8:14])
            print('%3dch %s  %.0f us, %d items; stamps in kilo-cycles of the shader clock' % (c, name, e0.elapsed_time(e1) * 1e3, v[6]))
            print('   compute wave: ' + ' | '.join('%s %.1f' % (n, x / 1000.0) for n, x in zip(CN, v[0:6])) + ' | total %.1f' % (ct / 1000.0))
            print('   memory wave : ' + ' | '.join('%s %.1f' % (n, x / 1000.0) for n, x in zip(MN, v[8:14])) + ' | total %.1f' % (mt / 1000.0), flush=True)
