"""-m gpu: temporal-convolution HIP kernel (forward, data gradient, fused BN/ReLU prologue, mask epilogue)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from gpu_util import dev, to_ntvc, to_nctv, diag

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 3e-5, torch.bfloat16: 2e-2, torch.float16: 5e-3}


@pytest.fixture(scope='module')
def ops():
    from istgcn_amd import ops as o
    return o


CASES = [
    # NM, Cin, Cout, T, V, k, stride
    (2, 64, 64, 23, 25, 9, 1), (2, 64, 128, 20, 25, 9, 2), (1, 128, 128, 31, 25, 15, 1), (2, 256, 256, 12, 18, 9, 1),
    (2, 16, 16, 16, 25, 3, 1), (2, 8, 8, 17, 25, 15, 2), (3, 11, 11, 9, 25, 9, 1), (1, 64, 64, 300, 25, 9, 1),
    (2, 128, 256, 30, 25, 15, 2),
    # stride 2 with an odd input length and several tiles per sequence
    (3, 64, 128, 61, 25, 9, 2), (2, 128, 256, 50, 25, 9, 2),
]


def _mk(case, dt, seed=0):
    NM, cin, cout, T, V, k, s = case
    g = torch.Generator().manual_seed(1000 + seed + hash(case) % 1000)
    x = torch.randn(NM, cin, T, V, generator=g)
    W = torch.randn(cout, cin, k, 1, generator=g) * (cin * k) ** -0.5
    b = torch.randn(cout, generator=g) * 0.1
    sc = 0.5 + torch.rand(cin, generator=g)
    sh = torch.randn(cin, generator=g) * 0.3
    if dt != torch.float32:
        x = x.to(dt).float()
    return x, W, b, sc, sh


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', CASES)
def test_tconv_forward(ops, case, dt):
    NM, cin, cout, T, V, k, s = case
    x, W, b, sc, sh = _mk(case, dt)
    u = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if dt != torch.float32:
        u = u.to(dt).float()
    ref = F.conv2d(u, W, b, stride=(s, 1), padding=((k - 1) // 2, 0))
    Tout = ref.shape[2]
    d = dev()
    taps, in_mul = ops.conv_taps_fwd(k, s)
    wf = W[:, :, :, 0].permute(2, 0, 1).contiguous().to(d)          # [k][Cout][Cin]
    wp = ops.pack_tconv_weight(wf, V, taps, in_mul, dt)
    stats = torch.zeros(ops.STATS_REP, 2, cout, dtype=torch.float64, device=d)
    y = ops.tconv(to_ntvc(x).to(d, dt), wp, cout, taps, bias=b.to(d), pre=torch.stack([sc, sh]).to(d), pre_relu=True,
                  stats=stats, Tout=Tout, Mlog=Tout, in_mul=in_mul)
    torch.cuda.synchronize()
    name = 'tconv_fwd_%s_%s' % ('x'.join(map(str, case)), str(dt)[6:])
    assert diag(name, to_nctv(y.float()), ref, TOL[dt]) < TOL[dt]
    yf = y.double().cpu()
    s_ = stats.sum(0).cpu()
    assert rel_err(s_[0], yf.sum((0, 1, 2))) < 1e-5 and rel_err(s_[1], (yf * yf).sum((0, 1, 2))) < 1e-5


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', CASES)
def test_tconv_data_gradient_with_mask(ops, case, dt):
    """du = conv^T(dz) per output phase, then the ReLU mask of the producer BN recomputed from `aux` and the two
    BatchNorm-backward reductions sum(d1), sum(d1 * xhat)."""
    NM, cin, cout, T, V, k, s = case
    x, W, b, sc, sh = _mk(case, dt, seed=1)
    g = torch.Generator().manual_seed(77)
    Tz = (T + 2 * ((k - 1) // 2) - k) // s + 1
    dz = torch.randn(NM, cout, Tz, V, generator=g)
    mean = torch.randn(cin, generator=g) * 0.2
    rstd = 0.5 + torch.rand(cin, generator=g)
    if dt != torch.float32:
        dz = dz.to(dt).float()
    u = x.clone().requires_grad_(True)
    F.conv2d(u, W, None, stride=(s, 1), padding=((k - 1) // 2, 0)).backward(dz)
    on = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0
    ref = u.grad * on
    d = dev()
    out = torch.full((NM, T, V, cin), float('nan'), device=d, dtype=dt)
    stats = torch.zeros(ops.STATS_REP, 2, cin, dtype=torch.float64, device=d)
    maux = torch.stack([sc, sh, mean, rstd]).to(d)
    xg = to_ntvc(x).to(d, dt)
    for phase in range(s):
        tl = ops.conv_taps_bwd(k, s, phase)
        Mlog = (T - phase + s - 1) // s
        if not tl:
            continue
        wt = torch.stack([W[:, :, j, 0].t() for j, _ in tl]).contiguous().to(d)      # [taps][Cin][Cout]
        offs = [dj for _, dj in tl]
        wp = ops.pack_tconv_weight(wt, V, offs, 1, dt)
        ops.tconv(to_ntvc(dz).to(d, dt), wp, cin, offs, aux=xg, maux=maux, out=out, stats=stats, mode=1,
                  Tout=T, Mlog=Mlog, in_mul=1, out_mul=s, out_off=phase)
    torch.cuda.synchronize()
    name = 'tconv_bwd_%s_%s' % ('x'.join(map(str, case)), str(dt)[6:])
    assert diag(name, to_nctv(out.float()), ref, TOL[dt]) < TOL[dt]
    of = out.double().cpu()
    xhat = (to_ntvc(x).double() - mean.double()) * rstd.double()
    s_ = stats.sum(0).cpu()
    assert rel_err(s_[0], of.sum((0, 1, 2))) < 1e-5
    assert rel_err(s_[1], (of * xhat).sum((0, 1, 2))) < 1e-4


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', CASES)
def test_tconv_weight_gradient(ops, case, dt):
    """dW[j][o][i] and dbias against autograd of relu(bn(x)) -> conv2d."""
    NM, cin, cout, T, V, k, s = case
    x, W, b, sc, sh = _mk(case, dt, seed=2)
    g = torch.Generator().manual_seed(78)
    u = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if dt != torch.float32:
        u = u.to(dt).float()
    Wp = W.clone().requires_grad_(True)
    bp = b.clone().requires_grad_(True)
    z = F.conv2d(u, Wp, bp, stride=(s, 1), padding=((k - 1) // 2, 0))
    dz = torch.randn(z.shape, generator=g)
    if dt != torch.float32:
        dz = dz.to(dt).float()
    z.backward(dz)
    d = dev()
    taps, in_mul = ops.conv_taps_fwd(k, s)
    dW, db = ops.tconv_wgrad(to_ntvc(dz).to(d, dt), to_ntvc(x).to(d, dt), taps, in_mul=in_mul,
                             pre=torch.stack([sc, sh]).to(d), pre_relu=True)
    torch.cuda.synchronize()
    ref = Wp.grad[:, :, :, 0].permute(2, 0, 1)
    tol = 2e-5 if dt == torch.float32 else 5e-3
    name = 'tconv_wgrad_%s_%s' % ('x'.join(map(str, case)), str(dt)[6:])
    assert diag(name, dW, ref, tol) < tol
    assert diag(name + '_db', db, bp.grad, tol) < tol


# the lean weight-gradient kernel (tconv_wgrad_lean.hip; taken when no bias gradient is asked for): NM, Cin, Cout, T, V, k and
# the grid cap (0 = resident size; a small one gives each workgroup a long walk: sliding windows, a region pass ending in
# a fresh window, sequence starts in the middle of a walk, the "late" fresh window right after a sequence start)
LEAN_WG_CASES = [
    ((2, 64, 64, 23, 25, 9), 0), ((1, 64, 64, 300, 25, 9), 2), ((3, 128, 64, 41, 25, 9), 4), ((5, 64, 128, 7, 25, 9), 2),
    ((2, 256, 256, 12, 18, 9), 0), ((2, 256, 256, 40, 18, 9), 16), ((2, 64, 128, 37, 25, 5), 2), ((1, 64, 64, 50, 20, 7), 1),
    ((2, 64, 64, 3, 25, 9), 0), ((4, 64, 64, 5, 25, 9), 1), ((1, 128, 128, 151, 25, 9), 4),
    # 10..15 taps: two launches (8 + the rest), each with its own window
    ((1, 128, 128, 31, 25, 15), 0), ((2, 64, 64, 40, 25, 15), 2), ((2, 64, 128, 20, 25, 11), 1), ((2, 128, 64, 33, 18, 15), 2),
    # stride 2 (a 7th entry): one unit-stride launch per tap parity on every second frame; odd and even input lengths
    ((2, 64, 128, 20, 25, 9, 2), 0), ((3, 64, 128, 61, 25, 9, 2), 2), ((2, 128, 256, 50, 25, 9, 2), 4), ((1, 128, 128, 300, 25, 9, 2), 2),
    ((2, 64, 64, 33, 18, 9, 2), 1), ((4, 64, 64, 7, 25, 9, 2), 1),
    ((2, 64, 128, 41, 25, 15, 2), 2), ((1, 128, 128, 60, 25, 15, 2), 0),
    # 1 and 3 taps (the 1 x 1 residual conv at stride 2; windows without / with two shared frames)
    ((2, 64, 128, 61, 25, 1, 2), 2), ((3, 128, 256, 20, 25, 1, 2), 4), ((2, 64, 64, 37, 25, 1, 1), 1), ((2, 64, 128, 44, 25, 3, 1), 2),
    ((6, 64, 64, 9, 25, 4, 1), 1), ((5, 64, 64, 11, 18, 4, 1), 1),
    # the second launch of a 15-tap layer through the WORKSPACE (>= 128 slices: its taps are local there, 8 .. 14 in dW), and a
    # long walk over several sequences at stride 2 with 8 + 7 taps per parity
    ((40, 64, 64, 20, 25, 15), 0), ((4, 256, 256, 150, 25, 15, 2), 1), ((3, 128, 128, 150, 25, 15, 2), 1),
    # one tile per sequence (every window fresh and "late"), more workgroups than tile chunks (a workgroup without tiles)
    ((5, 64, 64, 5, 25, 9), 4), ((7, 128, 64, 4, 25, 9, 2), 4),
]


@pytest.mark.parametrize('with_pre', [True, False])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case,cap', LEAN_WG_CASES)
def test_tconv_weight_gradient_without_bias(ops, case, cap, dt, with_pre):
    """dW alone (the training step's call: the bias feeds a batch-statistics BatchNorm) against autograd of
    relu(bn(x)) -> conv2d, at shapes the lean kernel serves; the bias buffer handed in must come back untouched."""
    stride = case[6] if len(case) > 6 else 1
    NM, cin, cout, T, V, k = case[:6]
    x, W, b, sc, sh = _mk(tuple(case[:6]) + (stride,), dt, seed=3)
    g = torch.Generator().manual_seed(79)
    u = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if with_pre else x
    if dt != torch.float32:
        u = u.to(dt).float()
    Wp = W.clone().requires_grad_(True)
    z = F.conv2d(u, Wp, None, stride=(stride, 1), padding=((k - 1) // 2, 0))
    dz = torch.randn(z.shape, generator=g)
    dz = dz.to(dt).float()
    z.backward(dz)
    d = dev()
    taps, in_mul = ops.conv_taps_fwd(k, stride)
    blocks = (cin // 64) * (cout // 64)
    out = (torch.zeros(k, cout, cin, device=d), torch.full((cout,), 7.0, device=d))
    dW, db = ops.tconv_wgrad(to_ntvc(dz).to(d, dt), to_ntvc(x).to(d, dt), taps, in_mul=in_mul,
                             pre=torch.stack([sc, sh]).to(d) if with_pre else None, pre_relu=with_pre, want_bias=False,
                             grid_cap=cap * blocks, out=out)
    torch.cuda.synchronize()
    ref = Wp.grad[:, :, :, 0].permute(2, 0, 1)
    name = 'tconv_wgrad_nobias_%s_cap%d_%s_%s' % ('x'.join(map(str, case)), cap, str(dt)[6:], 'pre' if with_pre else 'raw')
    assert diag(name, dW, ref, 5e-3) < 5e-3
    assert bool((db == 7.0).all())


@pytest.mark.parametrize('k,s', [(1, 2), (3, 4), (1, 1)])
def test_conv_data_gradient_with_empty_phases(ops, k, s):
    """k < stride: some output phases of the data gradient receive no tap.  Those frames must be zero and the OTHER
    phases must keep what their launches wrote (a whole-tensor zero_() in a later phase wiped them once)."""
    from istgcn_amd import functional as Fn
    NM, C, T, V = 2, 16, 13, 25
    g = torch.Generator().manual_seed(5)
    W = torch.randn(C, C, k, 1, generator=g) * (C * k) ** -0.5
    x = torch.randn(NM, C, T, V, generator=g).requires_grad_(True)
    z = F.conv2d(x, W, None, stride=(s, 1), padding=((k - 1) // 2, 0))
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    d = dev()
    taps = W[:, :, :, 0].permute(2, 0, 1).contiguous().to(d)       # [k][Cout][Cin]
    dx = Fn._conv_bwd_data(to_ntvc(dz).to(d), taps, k, s, T, C, V)
    assert diag('conv_bwd_empty_phase_k%d_s%d' % (k, s), to_nctv(dx), x.grad, 3e-5) < 3e-5
    assert float(x.grad.abs().max()) > 0
