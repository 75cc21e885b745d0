"""Diagnostic: per-phase cycle shares of the tconv kernel (needs a library built with -DISTGCN_STAMP)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import istgcn_amd
from istgcn_amd import ops, _lib
lib = _lib.load()
d = torch.device('cuda:0'); dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
for c, T in ((64, 300), (128, 150), (256, 75)):
    NM, V, k = 128, 25, 9
    g = torch.randn(NM, T, V, c, device=d).to(dt)
    taps, im = ops.conv_taps_fwd(k, 1)
    wp = ops.pack_tconv_weight(torch.randn(k, c, c, device=d) * 0.05, V, taps, im, dt)
    pre = torch.stack([torch.ones(c), torch.zeros(c)]).to(d); bias = torch.zeros(c, device=d); st = ops.new_stats(c, d)
    f = lambda: ops.tconv(g, wp, c, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1)
    f(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.istgcn_debug_stamps(out, 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    lib.istgcn_debug_stamps(out, 1)
    v = list(out); tot = sum(v[:6]) or 1
    names = ['tile setup', 'staging (loads+transform+LDS write)', 'barrier after staging', 'MFMA loop', 'barrier after MFMA', 'epilogue']
    print('C=%d: ' % c + ', '.join('%s %.1f%%' % (n, 100 * x / tot) for n, x in zip(names, v[:6])), flush=True)
