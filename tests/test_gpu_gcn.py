"""-m gpu: the graph-conv HIP kernel (through the C ABI) against the golden fixtures and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from gpu_util import dev, to_ntvc, to_nctv, diag, OUT, sub_close
from detinit import WIDE_UNITS, wide_unit_inputs, wide_unit_names
from oracle import stgcn_ref as R

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 5e-3}


@pytest.fixture(scope='module')
def ops():
    from istgcn_amd import ops as o
    return o


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_probe_mfma_lane_maps(ops, dt):
    """A = exact small integers, asymmetric B: pins the A/B/D lane maps of common.hpp."""
    import ctypes
    kgs = 8 if dt == torch.float32 else 16
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-4, 5, (32, kgs), generator=g).float()
    Bt = torch.randint(-4, 5, (32, kgs), generator=g).float() + torch.arange(32).float()[:, None] * 0.0
    Bt[:, 0] += torch.arange(32).float()        # asymmetric
    D = torch.zeros(32, 32, device=dev())
    a, b = A.to(dev(), dt).contiguous(), Bt.to(dev(), dt).contiguous()
    ops._call('istgcn_probe_mfma', ops._ptr(a), ops._ptr(b), ops._ptr(D), ops.dtype_code(a), ops._stream(a))
    torch.cuda.synchronize()
    assert torch.equal(D.cpu(), A @ Bt.t())


def test_probe_tr16(ops):
    """ds_read_b64_tr_b16: record what each lane receives for the addressing the bf16 kernels use."""
    R_, C_ = 16, 32
    src = (torch.arange(R_)[:, None] * 256 + torch.arange(C_)[None, :]).to(torch.int16).contiguous()
    lane = torch.arange(64)
    grp, idx = lane // 16, lane % 16
    q, p = idx // 4, idx % 4
    # group g: block rows 4*(g>>1).. , columns 16*(g&1)..
    row = 4 * (grp // 2) + q
    col = 16 * (grp % 2) + 4 * p
    off = ((row * C_ + col) * 2).to(torch.int32)
    out = torch.zeros(64, 4, dtype=torch.int16, device=dev())
    s, o = src.to(dev()), off.to(dev())
    ops._call('istgcn_probe_tr16', ops._ptr(s), R_ * C_, ops._ptr(o), ops._ptr(out), ops._stream(s))
    torch.cuda.synchronize()
    got = out.cpu().to(torch.int32) & 0xFFFF
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'probe_tr16.txt'), 'w') as f:
        for l in range(64):
            f.write('lane %2d addr(row %2d col %2d) -> %s\n' % (l, row[l], col[l], ['r%dc%d' % (v >> 8, v & 255) for v in got[l].tolist()]))
    # expectation (cdna guide T10): lane i of a 16-lane group receives column i of the 4 rows
    exp = torch.stack([(4 * (grp // 2) + j) * 256 + 16 * (grp % 2) + idx for j in range(4)], 1)
    assert torch.equal(got, exp.to(torch.int32))


def _unit(golden, ci):
    g = golden('units_g2.npz')
    b = 'c%d.' % ci
    return g, b, (lambda k: torch.from_numpy(g[b + k]))


def _fold(unit, t):
    A, A2, A3 = t('A'), t('A2'), t('A3')
    i1, i2, i3 = t('imp1'), t('imp2'), t('imp3')
    if unit == 'tgcn':
        return A * i1
    if unit == '3a':
        return A * i1 + A ** 2 * i2 + A ** 3 * i3
    if unit in ('inc', 'incnew'):
        return A * i1 + A2 * i2 + A3 * i3
    Ai = A * i1
    return {'multi3': Ai + Ai ** 2 + Ai ** 3, 'multi3fix': (Ai + Ai ** 2 + Ai ** 3) / 3, 'only3': Ai ** 3}[unit]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('unit', ['tgcn', '3a', 'inc', 'incnew', 'multi3', 'multi3fix', 'only3'])
def test_gcn_unit_golden(ops, golden, unit, ci, dt):
    """y and dx of every GCN-unit variant of the reference = ONE kernel with a folded adjacency."""
    g, b, t = _unit(golden, ci)
    x, r, W, bias = t('x'), t('r'), t('W'), t('b')
    Aeff = _fold(unit, t)
    K, V = Aeff.shape[0], Aeff.shape[1]
    cout, cin = W.shape[0] // K, W.shape[1]
    d = dev()
    Wk = W.view(K, cout, cin)
    wr = Wk.permute(1, 0, 2).contiguous().to(d)                     # Wr[c][k][i]
    bterm = torch.einsum('kc,kw->wc', bias.view(K, cout), Aeff.sum(1)).contiguous().to(d)
    xg = to_ntvc(x).to(d, dt)
    y = ops.gcn_forward(xg, Aeff.to(d).contiguous(), ops.pack_gcn_weight(wr, dt), cout, bterm=bterm)
    torch.cuda.synchronize()
    name = 'gcn_%s_c%d_%s' % (unit, ci, str(dt)[6:])
    assert diag(name + '_y', to_nctv(y.float()), g[b + unit + '.y'], TOL[dt]) < TOL[dt]
    # data gradient: same kernel, A^T and Wr^T  (W'[i][k][c] = W[k*Cout+c][i])
    wt = Wk.permute(2, 0, 1).contiguous().to(d)
    dy = to_ntvc(r).to(d, dt)
    dx = ops.gcn_forward(dy, Aeff.transpose(1, 2).contiguous().to(d), ops.pack_gcn_weight(wt, dt), cin)
    torch.cuda.synchronize()
    assert diag(name + '_dx', to_nctv(dx.float()), g[b + unit + '.dx'], TOL[dt]) < TOL[dt]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # NM, Cin, Cout, T, V, K, dense
    (3, 64, 64, 23, 25, 3, False), (2, 64, 128, 11, 25, 3, False), (2, 128, 256, 9, 25, 3, False),
    (2, 256, 256, 7, 18, 3, True), (1, 3, 64, 300, 25, 3, False), (2, 64, 64, 12, 25, 4, True),
    (2, 40, 24, 5, 15, 2, True), (5, 96, 320, 6, 19, 1, True),
])
def test_gcn_fwd_random_vs_oracle(ops, shape, dt):
    """ragged tiles, channel padding, K=1..4, dense adjacency, BatchNorm partial sums, addend."""
    NM, cin, cout, T, V, K, dense = shape
    g = torch.Generator().manual_seed(hash(shape) & 0xFFFF)
    x = torch.randn(NM, cin, T, V, generator=g)
    W = torch.randn(K * cout, cin, 1, 1, generator=g) * cin ** -0.5
    bias = torch.randn(K * cout, generator=g) * 0.1
    A = torch.rand(K, V, V, generator=g)
    if not dense:
        A = A * (torch.rand(K, V, V, generator=g) < 0.12)
    add = torch.randn(NM, cout, T, V, generator=g)
    if dt != torch.float32:   # compare like with like: oracle sees the same rounded inputs
        x, add = x.to(dt).float(), add.to(dt).float()
    ref = R.graph_einsum(torch.nn.functional.conv2d(x, W, bias), A) + add
    d = dev()
    wr = W.view(K, cout, cin).permute(1, 0, 2).contiguous().to(d)
    bterm = torch.einsum('kc,kw->wc', bias.view(K, cout), A.sum(1)).contiguous().to(d)
    stats = torch.zeros(ops.STATS_REP, 2, cout, dtype=torch.float64, device=d)
    y = ops.gcn_forward(to_ntvc(x).to(d, dt), A.to(d), ops.pack_gcn_weight(wr, dt), cout, bterm=bterm,
                        addend=to_ntvc(add).to(d, dt), stats=stats, nnz_cap=int((A != 0).sum()))
    torch.cuda.synchronize()
    name = 'gcnrand_%s_%s' % ('x'.join(map(str, shape[:6])), str(dt)[6:])
    assert diag(name, to_nctv(y.float()), ref, TOL[dt]) < TOL[dt]
    yf = y.double().cpu()
    s = stats.sum(0).cpu()
    assert rel_err(s[0], yf.sum((0, 1, 2))) < 1e-5
    assert rel_err(s[1], (yf * yf).sum((0, 1, 2))) < 1e-5


# ------------------------------------------------------------------------------------------------ G2W: the bench widths
def _wide_graph(V):
    from istgcn_amd.net.utils.graph import Graph
    g = Graph('ntu-rgb+d' if V == 25 else 'openpose', 'spatial_3')
    return tuple(torch.tensor(a, dtype=torch.float32) for a in (g.A, g.A2, g.A3))


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('ci', range(len(WIDE_UNITS)))
def test_gcn_unit_wide_golden(ops, golden, ci, dt):
    """the register-chained graph-conv kernels (csrc/gcn_rc*.hip: 64 / 128 / 256 channels) against
    REFERENCE-generated outputs and gradients (units_g2w.npz: net/utils/tgcn.py:76-89, tgcn_multi3_fix_3A.py:76-92,
    inceptionv2_gcn.py:64-89 run at (64,64), (64,128), (128,256) channels), all three storage types: y, dx, dW, db and the
    importance gradients, with the check that the register-chained entries were the ones dispatched."""
    import ctypes
    from istgcn_amd import _lib
    g = golden('units_g2w.npz')
    cin, cout, V = WIDE_UNITS[ci]
    A, A2, A3 = _wide_graph(V)
    K = A.shape[0]
    x, r, W, bias = wide_unit_inputs(ci, K)
    b = 'w%d.' % ci
    lib = _lib.load()
    # the dispatchers' own predicates (gcn_fwd.hip / gcn_bwd.hip / tconv_wgrad.hip): 16-bit -> gcn_rc*, fp32 -> gcn_rc_f32 for V >= 20
    if dt != torch.float32 or V >= 20:
        assert lib.istgcn_gcn_rc_layout(cin, cout, K, ops._DT[dt]) == 1
        assert lib.istgcn_gcn_wgrad_rc_ok(V, cin, cout, K, ops._DT[dt]) == 1 or dt == torch.float32
    d = dev()
    tol_y = TOL[dt]
    tol_g = 3e-5 if dt == torch.float32 else 1e-2
    for unit in wide_unit_names(ci):
        imps = [torch.from_numpy(g[b + 'imp%d' % j]).clone().requires_grad_(True) for j in (1, 2, 3)]
        if unit == 'tgcn':
            Aeff = A * imps[0]
        elif unit == '3a':
            Aeff = A * imps[0] + A ** 2 * imps[1] + A ** 3 * imps[2]
        elif unit in ('inc', 'incnew'):
            Aeff = A * imps[0] + A2 * imps[1] + A3 * imps[2]
        else:
            Ai = A * imps[0]
            Aeff = {'multi3': Ai + Ai ** 2 + Ai ** 3, 'multi3fix': (Ai + Ai ** 2 + Ai ** 3) / 3, 'only3': Ai ** 3}[unit]
        Ad = Aeff.detach().to(d).contiguous()
        cap = int((Aeff != 0).sum())
        Wk = W.view(K, cout, cin)
        wr = Wk.permute(1, 0, 2).contiguous().to(d)
        bterm = torch.einsum('kc,kw->wc', bias.view(K, cout), Aeff.detach().sum(1)).contiguous().to(d)
        xg, dyg = to_ntvc(x).to(d, dt), to_ntvc(r).to(d, dt)
        name = 'gcnw_%s_w%d_%s' % (unit, ci, str(dt)[6:])
        k = b + unit
        y = ops.gcn_forward(xg, Ad, ops.pack_gcn_weight(wr, dt), cout, bterm=bterm, nnz_cap=cap)
        assert sub_close(name + '_y', to_nctv(y.float()), g, k + '.y', tol_y if dt == torch.float32 else 1e-2, dt), unit
        dW, S = ops.gcn_wgrad(dyg, xg, Ad, nnz_cap=cap)
        dx, dA = ops.gcn_bwd_data(dyg, Ad, Wk.contiguous().to(d), x=xg, nnz_cap=cap)
        torch.cuda.synchronize()
        assert sub_close(name + '_dx', to_nctv(dx.float()), g, k + '.dx', tol_y if dt == torch.float32 else 1e-2, dt), unit
        dW, dA, S = dW.cpu(), dA.cpu(), S.cpu()
        assert sub_close(name + '_dW', dW.view(K * cout, cin, 1, 1), g, k + '.dW', tol_g, dt), unit
        db = torch.einsum('kw,wc->kc', Aeff.detach().sum(1), S).reshape(-1)
        assert sub_close(name + '_db', db, g, k + '.db', tol_g, dt), unit
        dA_full = dA + torch.einsum('kc,wc->kw', bias.view(K, cout), S)[:, None, :] * (Aeff.detach() != 0)
        Aeff.backward(dA_full)
        for j in (1, 2, 3):
            if k + '.dimp%d' % j in g.files:
                assert sub_close(name + '_dimp%d' % j, imps[j - 1].grad, g, k + '.dimp%d' % j, tol_g, dt), (unit, j)


def test_gcn_strided_and_accumulate(ops):
    """K=1, A=I, in stride 2 = the residual Conv2d(1x1, stride (2,1)) of st_gcnold.py:186-191; its data
    gradient scatters into every other frame of an existing tensor (accumulate + out stride)."""
    g = torch.Generator().manual_seed(9)
    NM, cin, cout, T, V = 2, 64, 128, 14, 25
    x = torch.randn(NM, cin, T, V, generator=g)
    W = torch.randn(cout, cin, 1, 1, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x, W, bias, stride=(2, 1))
    d = dev()
    eye = torch.eye(V).view(1, V, V).contiguous().to(d)
    bterm = bias.view(1, cout).expand(V, cout).contiguous().to(d)
    y = ops.gcn_forward(to_ntvc(x).to(d), eye, ops.pack_gcn_weight(W.view(cout, 1, cin).to(d), torch.float32), cout,
                        bterm=bterm, in_t_stride=2, nnz_cap=V)
    assert diag('gcn_strided', to_nctv(y), ref, 2e-5) < 2e-5
    dy = torch.randn(NM, cout, T // 2, V, generator=g)
    base = torch.randn(NM, cin, T, V, generator=g)
    xx = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xx, W, bias, stride=(2, 1)).backward(dy)
    buf = to_ntvc(base).to(d)
    wt = W.view(cout, cin).t().contiguous().view(cin, 1, cout).to(d)
    ops.gcn_forward(to_ntvc(dy).to(d), eye, ops.pack_gcn_weight(wt, torch.float32), cin, addend=buf, out=buf,
                    Tout=T, out_t_stride=2, nnz_cap=V)
    assert diag('gcn_strided_bwd', to_nctv(buf), base + xx.grad, 2e-5) < 2e-5


def test_requires_gpu_no_fallback(ops):
    x = torch.zeros(1, 2, 25, 8)
    with pytest.raises(RuntimeError):
        ops.gcn_forward(x, torch.zeros(1, 25, 25), torch.zeros(8), 8)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('unit', ['tgcn', '3a', 'inc'])
def test_gcn_param_grads_golden(ops, golden, unit, ci, dt):
    """dW, db and the importance gradients of the reference units from ONE wgrad kernel: the kernel returns dW, the
    adjacency gradient on the pattern and S = sum dy; the fold A_eff(importances) and the bias term are differentiated
    by the same tiny host expressions the Model uses."""
    g, b, t = _unit(golden, ci)
    x, r, W, bias = t('x'), t('r'), t('W'), t('b')
    A, A2, A3 = t('A'), t('A2'), t('A3')
    imps = [t('imp%d' % j).clone().requires_grad_(True) for j in (1, 2, 3)]
    if unit == 'tgcn':
        Aeff = A * imps[0]
    elif unit == '3a':
        Aeff = A * imps[0] + A ** 2 * imps[1] + A ** 3 * imps[2]
    else:
        Aeff = A * imps[0] + A2 * imps[1] + A3 * imps[2]
    K, V = A.shape[0], A.shape[1]
    cout, cin = W.shape[0] // K, W.shape[1]
    d = dev()
    xin, rin = x, r
    if dt != torch.float32:
        xin, rin = x.to(dt).float(), r.to(dt).float()
    w3 = W.view(K, cout, cin).contiguous().to(d)
    cap = int((Aeff != 0).sum())
    dyg, xg, Ag = to_ntvc(rin).to(d, dt), to_ntvc(xin).to(d, dt), Aeff.detach().to(d).contiguous()
    dW, S = ops.gcn_wgrad(dyg, xg, Ag, nnz_cap=cap)
    dx2, dA = ops.gcn_bwd_data(dyg, Ag, w3, x=xg, nnz_cap=cap)
    torch.cuda.synchronize()
    dW, dA, S = dW.cpu(), dA.cpu(), S.cpu()
    assert diag('gcnbwd_%s_c%d_%s_dx' % (unit, ci, str(dt)[6:]), to_nctv(dx2.float()), g[b + unit + '.dx'], TOL[dt]) < TOL[dt]
    tol = 3e-5 if dt == torch.float32 else 1e-2
    name = 'gcnwg_%s_c%d_%s' % (unit, ci, str(dt)[6:])
    assert diag(name + '_dW', dW.view(K * cout, cin, 1, 1), g[b + unit + '.dW'], tol) < tol
    # db[k*C+c] = sum_w colsum_k(A)[w] * S[w][c]
    db = torch.einsum('kw,wc->kc', Aeff.detach().sum(1), S).reshape(-1)
    assert diag(name + '_db', db, g[b + unit + '.db'], tol) < tol
    # adjacency gradient: kernel part (through W x) + bias part (every v of column w gets sum_c b[k,c] S[w][c])
    dA_full = dA + torch.einsum('kc,wc->kw', bias.view(K, cout), S)[:, None, :] * (Aeff.detach() != 0)
    Aeff.backward(dA_full)
    for j in (1, 2, 3):
        key = b + unit + '.dimp%d' % j
        if key in g.files:
            assert diag(name + '_dimp%d' % j, imps[j - 1].grad, g[key], tol) < tol


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 64, 64, 23, 25, 3, False), (2, 64, 128, 11, 25, 3, False), (2, 128, 256, 9, 25, 3, False),
                                   (2, 256, 256, 7, 18, 3, True), (1, 3, 64, 300, 25, 3, False), (2, 64, 64, 12, 25, 4, True),
                                   (2, 40, 24, 5, 15, 2, True),
                                   # more tiles than workgroups (a workgroup walks several tiles); two input chunks with
                                   # one output chunk (the wave-specialised gcn_bwd_data's non-resident weight path)
                                   (40, 64, 64, 38, 25, 3, False), (12, 128, 64, 150, 25, 3, False)])
def test_gcn_wgrad_random_vs_autograd(ops, shape, dt):
    NM, cin, cout, T, V, K, dense = shape
    gen = torch.Generator().manual_seed(hash(shape) & 0xFFFF)
    x = torch.randn(NM, cin, T, V, generator=gen)
    dy = torch.randn(NM, cout, T, V, generator=gen)
    if dt != torch.float32:
        x, dy = x.to(dt).float(), dy.to(dt).float()
    W = (torch.randn(K * cout, cin, 1, 1, generator=gen) * cin ** -0.5).requires_grad_(True)
    A = torch.rand(K, V, V, generator=gen)
    if not dense:
        A = A * (torch.rand(K, V, V, generator=gen) < 0.12)
    A = A.requires_grad_(True)
    y = R.graph_einsum(torch.nn.functional.conv2d(x, W), A)
    y.backward(dy)
    d = dev()
    cap = int((A != 0).sum())
    dyg, xg = to_ntvc(dy).to(d, dt), to_ntvc(x).to(d, dt)
    add = torch.randn(NM, cin, T, V, generator=gen)
    if dt != torch.float32:
        add = add.to(dt).float()
    dW, S = ops.gcn_wgrad(dyg, xg, A.detach().to(d), nnz_cap=cap)
    dx2, dA = ops.gcn_bwd_data(dyg, A.detach().to(d), W.detach().view(K, cout, cin).to(d), x=xg,
                               addend=to_ntvc(add).to(d, dt), nnz_cap=cap)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    R.graph_einsum(torch.nn.functional.conv2d(xr, W.detach()), A.detach()).backward(dy)
    assert diag('gcnbwdrand_%s_%s_dx' % ('x'.join(map(str, shape[:6])), str(dt)[6:]), to_nctv(dx2.float()), xr.grad + add,
                TOL[dt]) < TOL[dt]
    tol = 3e-5 if dt == torch.float32 else 1e-2
    name = 'gcnwgrand_%s_%s' % ('x'.join(map(str, shape[:6])), str(dt)[6:])
    assert diag(name + '_dW', dW.cpu().view_as(W), W.grad, tol) < tol
    assert diag(name + '_dA', dA.cpu(), A.grad * (A.detach() != 0), tol) < tol
    assert diag(name + '_S', S.cpu(), to_ntvc(dy).sum((0, 1)), tol) < tol


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('with_x,with_add', [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize('cin,cout', [(64, 64), (64, 256), (128, 128)])
def test_gcn_bwd_data_optional_operands(ops, cin, cout, with_x, with_add, dt):
    """The wave-specialised gcn_bwd_data issues the loads of absent operands (x / addend) from a dummy address to keep its
    load counts fixed: the results must not depend on them (autograd of net/utils/tgcn.py:79-86)."""
    NM, T, V, K = 3, 23, 25, 3
    gen = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(NM, cin, T, V, generator=gen).to(dt).float()
    dy = torch.randn(NM, cout, T, V, generator=gen).to(dt).float()
    add = torch.randn(NM, cin, T, V, generator=gen).to(dt).float()
    W = torch.randn(K * cout, cin, 1, 1, generator=gen) * cin ** -0.5
    A = (torch.rand(K, V, V, generator=gen) * (torch.rand(K, V, V, generator=gen) < 0.12)).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    R.graph_einsum(torch.nn.functional.conv2d(xr, W), A).backward(dy)
    d = dev()
    cap = int((A != 0).sum())
    dx, dA = ops.gcn_bwd_data(to_ntvc(dy).to(d, dt), A.detach().to(d), W.view(K, cout, cin).to(d),
                              x=to_ntvc(x).to(d, dt) if with_x else None,
                              addend=to_ntvc(add).to(d, dt) if with_add else None, want_dA=with_x, nnz_cap=cap)
    torch.cuda.synchronize()
    want = xr.grad + (add if with_add else 0)
    name = 'gcnbwdopt_%dx%d_%d%d_%s' % (cin, cout, with_x, with_add, str(dt)[6:])
    assert diag(name + '_dx', to_nctv(dx.float()), want, TOL[dt]) < TOL[dt]
    if with_x:
        assert diag(name + '_dA', dA.cpu(), A.grad * (A.detach() != 0), 1e-2) < 1e-2
    else:
        assert dA is None


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_first_layer_adjacency_gradient_without_dx(ops, dt):
    """The models' first layer (3 input channels, net/st_gcnold.py:44) needs the adjacency gradient but no input gradient:
    istgcn_gcn_bwd_data with dx = NULL (16-bit storage) must give the same dA as the full call and as autograd."""
    NM, cin, cout, T, V, K = 5, 3, 64, 37, 25, 3
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(NM, cin, T, V, generator=gen).to(dt).float().requires_grad_(True)
    dy = torch.randn(NM, cout, T, V, generator=gen).to(dt).float()
    W = (torch.randn(K * cout, cin, 1, 1, generator=gen) * cin ** -0.5).to(dt).float()
    A = (torch.rand(K, V, V, generator=gen) * (torch.rand(K, V, V, generator=gen) < 0.12)).requires_grad_(True)
    R.graph_einsum(torch.nn.functional.conv2d(x, W), A).backward(dy)
    d = dev()
    cap = int((A != 0).sum())
    args = (to_ntvc(dy).to(d, dt), A.detach().to(d), W.view(K, cout, cin).to(d))
    xd = to_ntvc(x.detach()).to(d, dt)
    with ops.trace() as tr:
        dx0, dA0 = ops.gcn_bwd_data(*args, x=xd, nnz_cap=cap)
        dx1, dA1 = ops.gcn_bwd_data(*args, x=xd, nnz_cap=cap, want_dx=False)
    torch.cuda.synchronize()
    assert dx0 is not None and dx1 is None
    # round 5: BOTH forms run the register-chained narrow kernel (dx is what data_bn's gradient flows through; the round-1
    # kernel served it until then), and dx equals autograd's on the same 16-bit operands to the storage type's rounding
    assert tr.ran('gcn_rc_bwd_kernel') and not tr.ran('_114gcn_bwd_kernel'), sorted(tr.kernels)
    ref_dx = to_ntvc(x.grad)
    e_dx = float((dx0.float().cpu() - ref_dx).norm() / ref_dx.norm())
    assert dx0.shape == xd.shape and e_dx < (6e-3 if dt == torch.bfloat16 else 8e-4), e_dx
    ref = A.grad * (A.detach() != 0)
    assert diag('first_layer_dA_full_%s' % str(dt)[6:], dA0.cpu(), ref, 1e-2) < 1e-2
    assert diag('first_layer_dA_only_%s' % str(dt)[6:], dA1.cpu(), ref, 1e-2) < 1e-2
    # float32 storage has no dx-less form: the request is ignored, dx is computed as before
    dx2, _ = ops.gcn_bwd_data(to_ntvc(dy).to(d), A.detach().to(d), W.view(K, cout, cin).to(d), x=to_ntvc(x).to(d),
                              nnz_cap=cap, want_dx=False)
    assert dx2 is not None


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 64, 64, 23, 25, 3), (2, 64, 128, 11, 25, 3), (2, 128, 256, 9, 25, 3), (2, 256, 256, 7, 18, 3),
                                   (2, 128, 128, 12, 25, 2), (40, 64, 64, 38, 25, 3)])
def test_gcn_fwd_float32_matrix_core_path_vs_oracle(ops, shape):
    """float32 storage on the register-chained kernel (csrc/gcn_rc_f32.hip: contraction AND aggregation on
    v_mfma_f32_32x32x2_f32, exact fp32 products): y and the BatchNorm partial sums against the oracle (net/utils/tgcn.py:79-86)."""
    NM, cin, cout, T, V, K = shape
    g = torch.Generator().manual_seed(hash(shape) & 0xFFFF)
    x = torch.randn(NM, cin, T, V, generator=g)
    W = torch.randn(K * cout, cin, 1, 1, generator=g) * cin ** -0.5
    bias = torch.randn(K * cout, generator=g) * 0.1
    A = torch.rand(K, V, V, generator=g) * (torch.rand(K, V, V, generator=g) < 0.15)
    ref = R.graph_einsum(torch.nn.functional.conv2d(x, W, bias), A)
    d = dev()
    wr = W.view(K, cout, cin).permute(1, 0, 2).contiguous().to(d)
    bterm = torch.einsum('kc,kw->wc', bias.view(K, cout), A.sum(1)).contiguous().to(d)
    stats = torch.zeros(ops.STATS_REP, 2, cout, dtype=torch.float64, device=d)
    y = ops.gcn_forward(to_ntvc(x).to(d), A.to(d), ops.pack_gcn_weight(wr, torch.float32), cout, bterm=bterm, stats=stats,
                        nnz_cap=int((A != 0).sum()))
    torch.cuda.synchronize()
    assert diag('gcnf32_%s' % 'x'.join(map(str, shape)), to_nctv(y), ref, 2e-5) < 2e-5
    yf, s = y.double().cpu(), stats.sum(0).cpu()
    assert rel_err(s[0], yf.sum((0, 1, 2))) < 1e-5
    assert rel_err(s[1], (yf * yf).sum((0, 1, 2))) < 1e-5
