"""Per-kernel micro-benchmark at the bench.py layer shapes (st_gcn_msgcn, batch 64 -> NM=128): times each C-ABI entry
point in isolation with HIP events and prints us/call, algorithmic TFLOP/s and GB/s.  Optimisation loop tool."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
from istgcn_amd.net.utils.graph import Graph

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--nm', type=int, default=128)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--only', default='')
ap.add_argument('--layers', default='64x64x300,128x128x150,256x256x75')
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
d = torch.device('cuda:0')
g = Graph('ntu-rgb+d', 'spatial_3')
A = torch.tensor(g.A + g.A2 + g.A3, dtype=torch.float32, device=d)
cap = int((A != 0).sum())
K, V = 3, 25


def timeit(name, fn, flops, nbytes):
    if args.only and args.only not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    print('%-34s %9.1f us  %8.1f TFLOP/s  %8.1f GB/s' % (name, us, flops / us / 1e6, nbytes / us / 1e3), flush=True)


for spec in args.layers.split(','):
    cin, cout, T = (int(v) for v in spec.split('x'))
    NM = args.nm
    es = 2 if dt == torch.bfloat16 else 4
    P = NM * T * V
    x = torch.randn(NM, T, V, cin, device=d).to(dt)
    dy = torch.randn(NM, T, V, cout, device=d).to(dt)
    W = torch.randn(K * cout, cin, device=d) * cin ** -0.5
    W3 = W.view(K, cout, cin)
    wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)
    bterm = torch.randn(V, cout, device=d)
    tag = '%dx%d T=%d ' % (cin, cout, T)
    st = ops.new_stats(cout, d)
    timeit(tag + 'gcn_fwd', lambda: ops.gcn_forward(x, A, wp, cout, bterm=bterm, stats=st, nnz_cap=cap),
           2.0 * P * cout * K * cin, P * (cin + cout) * es)
    timeit(tag + 'gcn_bwd_data', lambda: ops.gcn_bwd_data(dy, A, W3, x=x, addend=x if cin == cout else None, nnz_cap=cap),
           2.0 * P * cout * K * cin, P * (cout + 3 * cin) * es)
    timeit(tag + 'gcn_wgrad', lambda: ops.gcn_wgrad(dy, x, A, nnz_cap=cap), 2.0 * P * cout * K * cin, P * (cin + cout) * es)
    c = cout
    gten = torch.randn(NM, T, V, c, device=d).to(dt)
    for k in (9,):
        taps, im = ops.conv_taps_fwd(k, 1)
        wf = torch.randn(k, c, c, device=d) * (c * k) ** -0.5
        wpt = ops.pack_tconv_weight(wf, V, taps, im, dt)
        pre = torch.stack([torch.ones(c), torch.zeros(c)]).to(d)
        bias = torch.zeros(c, device=d)
        timeit(tag + 'tconv_fwd k=%d' % k, lambda: ops.tconv(gten, wpt, c, taps, bias=bias, pre=pre, pre_relu=True, stats=st, Tout=T, Mlog=T, in_mul=1),
               2.0 * P * c * c * k, 2 * P * c * es)
        maux = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(d)
        timeit(tag + 'tconv_bwd k=%d' % k, lambda: ops.tconv(gten, wpt, c, taps, aux=gten, maux=maux, stats=st, mode=1, Tout=T, Mlog=T, in_mul=1),
               2.0 * P * c * c * k, 3 * P * c * es)
        timeit(tag + 'tconv_wgrad k=%d' % k, lambda: ops.tconv_wgrad(gten, gten, taps, in_mul=1, pre=pre, pre_relu=True),
               2.0 * P * c * c * k, 2 * P * c * es)
    coef = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(d)
    abc = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c)]).to(d)
    timeit(tag + 'block_out_fwd', lambda: ops.block_out_fwd(gten, coef[:2].contiguous(), gten, None, 0.5, 1), 3.0 * P * c, 3 * P * c * es)
    timeit(tag + 'block_out_bwd', lambda: ops.block_out_bwd(gten, gten, gten, coef, None, None, 0.5, 1), 6.0 * P * c, 4 * P * c * es)
    _, rm = ops.block_out_fwd(gten, coef[:2].contiguous(), gten, None, 0.5, 1, want_mask=True)
    timeit(tag + 'block_out_bwd mask', lambda: ops.block_out_bwd(gten, None, gten, coef, None, None, 0.5, 1, relu_mask=rm), 6.0 * P * c, 3.06 * P * c * es)
    timeit(tag + 'block_out_bwd mask nodres', lambda: ops.block_out_bwd(gten, None, gten, coef, None, None, 0.5, 1, relu_mask=rm, want_dres=False), 6.0 * P * c, 2.06 * P * c * es)
    timeit(tag + 'block_out_bwd mask nodres p=0', lambda: ops.block_out_bwd(gten, None, gten, coef, None, None, 0.0, 1, relu_mask=rm, want_dres=False), 6.0 * P * c, 2.06 * P * c * es)
    timeit(tag + 'block_out_fwd p=0', lambda: ops.block_out_fwd(gten, coef[:2].contiguous(), gten, None, 0.0, 1), 3.0 * P * c, 3 * P * c * es)
    timeit(tag + 'affine2 (drop, mask)', lambda: ops.affine2(gten, gten, abc, 0.5, 1, relu_mask=rm), 4.0 * P * c, 3.06 * P * c * es)
    timeit(tag + 'affine2 (drop)', lambda: ops.affine2(gten, gten, abc, 0.5, 1), 4.0 * P * c, 3 * P * c * es)
    timeit(tag + 'affine2', lambda: ops.affine2(gten, gten, abc), 4.0 * P * c, 3 * P * c * es)
