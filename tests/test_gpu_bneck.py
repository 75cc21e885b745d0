"""-m gpu: the register-chained bottleneck kernels (csrc/bneck_rc.hip: istgcn_bneck_in / istgcn_bneck_out) through the C ABI
against a float64 torch restatement of the reference's chain (net/st_gcn_mstcn_1x1.py:174-213, 258-265: tcn_start ->
conv_1x1_start -> tcn_1/2/3 x mstcn_importance -> conv_1x1_end -> tcn_end sums) on the SAME 16-bit inputs, with the
narrow intermediate rounded to the storage type where the kernels store it; then the whole st_gcn block with the new
kernels against the same block on the generic temporal-conv kernels (the path the golden fixtures pin), and the
config-5 layer shape through size-independent properties.  Tolerances: one storage rounding of the result (2^-8 relative
for bfloat16, 2^-11 for float16) plus fp32 accumulation noise."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev

pytestmark = pytest.mark.gpu

DT = (torch.bfloat16, torch.float16)
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


@pytest.fixture(scope='module')
def ops():
    from istgcn_amd import ops as o
    return o


def _randn(*shape, seed, dt=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev(), dt)


def _close(got, ref, dt, what, extra=1.0):
    ref = ref.double()
    err = float((got.double() - ref).abs().max())
    bound = extra * (EPS[dt] * float(ref.abs().max()) + 1e-6)
    assert err <= bound, '%s: max abs err %.3e > %.3e' % (what, err, bound)


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('C,Wn,Wp', [(64, 8, 8), (128, 11, 16), (256, 16, 16), (64, 5, 8)])
@pytest.mark.parametrize('pre', [True, False])
def test_bneck_in_vs_torch(ops, dt, C, Wn, Wp, pre):
    rows = 3 * 25 * 7 + 13                                   # not a multiple of the 32-row tile
    x = _randn(rows, C, seed=1, dt=dt)
    W = _randn(Wn, C, seed=2, scale=C ** -0.5)
    b = _randn(Wn, seed=3)
    coef = torch.stack([_randn(C, seed=4).abs() + 0.5, _randn(C, seed=5)]).contiguous() if pre else None
    y = ops.bneck_in(x, W, Wp, bias=b, pre=coef, pre_relu=True)
    u = x.double()
    if pre:
        u = torch.relu(u * coef[0].double() + coef[1].double()).to(dt).double()     # the kernel rounds the transformed operand
    ref = u @ W.double().t() + b.double()
    assert y.shape == (rows, Wp)
    _close(y[:, :Wn], ref, dt, 'bneck_in', extra=1.5)
    assert float(y[:, Wn:].abs().max() if Wp > Wn else 0.0) == 0.0, 'padding channels must be exact zeros'
    # transposed weight view read in place (the backward use: dyb = We^T dz)
    Wt_ = _randn(C, Wn, seed=6, scale=C ** -0.5)
    y2 = ops.bneck_in(x, Wt_.t(), Wp)
    _close(y2[:, :Wn], x.double() @ Wt_.double(), dt, 'bneck_in (transposed view)', extra=1.5)


def _chain_ref(q, Wt, bt, We, be, taps_off, in_mul, Mlog, Wn, dt):
    """float64 restatement: yb = sum_j Wt_j q(in_mul m + off_j) + bt (rounded to dt), z = We yb + be."""
    NM, Tin, V, Wp = q.shape
    qd = q.double()[..., :Wn]
    yb = torch.zeros(NM, Mlog, V, Wn, dtype=torch.float64, device=q.device)
    for j, off in enumerate(taps_off):
        for m in range(Mlog):
            f = in_mul * m + off
            if 0 <= f < Tin:
                yb[:, m] += qd[:, f] @ Wt[j].double().t()
    if bt is not None:
        yb += bt.double()
    ybr = yb.to(dt).double()
    z = ybr @ We.double().t()
    if be is not None:
        z += be.double()
    return yb, z


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('C,Wn,Wp,V,stride', [(64, 8, 8, 25, 1), (128, 11, 16, 25, 2), (256, 16, 16, 18, 1), (64, 8, 8, 18, 2)])
def test_bneck_out_forward_vs_torch(ops, dt, C, Wn, Wp, V, stride):
    NM, T, k = 3, 41, 15
    Tz = (T - 1) // stride + 1
    q = _randn(NM, T, V, Wp, seed=1, dt=dt)
    q[..., Wn:] = 0
    Wt = _randn(k, Wn, Wn, seed=2, scale=(k * Wn) ** -0.5)
    bt, be = _randn(Wn, seed=3), _randn(C, seed=4)
    We = _randn(C, Wn, seed=5, scale=Wn ** -0.5)
    taps, in_mul = ops.conv_taps_fwd(k, stride)
    st = ops.new_stats(C, dev())
    yb, z = ops.bneck_out(q, Wt, list(range(k)), taps[0], We, C, bt=bt, be=be, stats=st, mode=0, Tout=Tz, Mlog=Tz, in_mul=in_mul)
    yb_ref, z_ref = _chain_ref(q, Wt, bt, We, be, taps, in_mul, Tz, Wn, dt)
    _close(yb[..., :Wn], yb_ref, dt, 'yb', extra=1.5)
    # z is computed from the ROUNDED yb the kernel stored: compare against the chain continued from the kernel's own yb
    z_from = yb[..., :Wn].double() @ We.double().t() + be.double()
    _close(z, z_from, dt, 'z', extra=1.5)
    _close(z, z_ref, dt, 'z vs chain', extra=4.0)
    if Wp > Wn:
        assert float(yb[..., Wn:].abs().max()) == 0.0
    # BatchNorm sums of the epilogue == sums of the stored tensor
    s = st.sum(0)
    zd = z.double().reshape(-1, C)
    assert torch.allclose(s[0], zd.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (zd * zd).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('C,Wn,Wp,stride', [(64, 8, 8, 1), (128, 11, 16, 2), (256, 16, 16, 1)])
def test_bneck_out_backward_mode_vs_torch(ops, dt, C, Wn, Wp, stride):
    """mode 1 with the data-gradient taps of a stride-`stride` 15-tap conv: dq = sum_j Wt_j^T dyb, d1 = mask * Ws^T dq, sums."""
    NM, T, V, k = 2, 37, 25, 15
    Tz = (T - 1) // stride + 1
    dyb = _randn(NM, Tz, V, Wp, seed=1, dt=dt)
    dyb[..., Wn:] = 0
    Wt = _randn(k, Wn, Wn, seed=2, scale=(k * Wn) ** -0.5)          # forward taps [k][n'][n]
    Ws = _randn(Wn, C, seed=3, scale=Wn ** -0.5)                    # forward [Wn][C]
    g = _randn(NM, T, V, C, seed=4, dt=dt)
    coef = torch.stack([_randn(C, seed=5).abs() + 0.5, _randn(C, seed=6) * 0.3, _randn(C, seed=7) * 0.1,
                        _randn(C, seed=8).abs() + 0.5]).contiguous()
    st = ops.new_stats(C, dev())
    dq = torch.zeros(NM, T, V, Wp, dtype=dt, device=dev())
    d1 = torch.zeros(NM, T, V, C, dtype=dt, device=dev())
    for phase in range(stride):
        tl = sorted(ops.conv_taps_bwd(k, stride, phase), key=lambda jd: jd[1])
        ops.bneck_out(dyb, Wt.transpose(1, 2), [j for j, _ in tl], tl[0][1], Ws.t(), C, aux=g, maux=coef, stats=st, mode=1,
                      Tout=T, Mlog=(T - phase + stride - 1) // stride, in_mul=1, out_mul=stride, out_off=phase, yb=dq, z=d1)
    # reference through autograd of the forward conv in float64
    qv = torch.zeros(NM, Wn, T, V, dtype=torch.float64, device=dev(), requires_grad=True)
    w4 = Wt.double().permute(1, 2, 0).unsqueeze(-1)                 # [n'][n][k][1]
    yv = F.conv2d(qv, w4, None, (stride, 1), ((k - 1) // 2, 0))
    yv.backward(dyb[..., :Wn].double().permute(0, 3, 1, 2))
    dq_ref = qv.grad.permute(0, 2, 3, 1)
    _close(dq[..., :Wn], dq_ref, dt, 'dq', extra=1.5)
    d1_lin = dq[..., :Wn].double() @ Ws.double()                   # from the kernel's own rounded dq
    mask = (g.double() * coef[0].double() + coef[1].double()) > 0
    # (positions within one rounding of the mask threshold may flip: compare where the decision is clear)
    clear = (g.double() * coef[0].double() + coef[1].double()).abs() > 1e-2
    ref = torch.where(mask, d1_lin, torch.zeros_like(d1_lin))
    err = ((d1.double() - ref).abs() * clear).max()
    assert float(err) <= 1.5 * (EPS[dt] * float(ref.abs().max()) + 1e-6)
    s = st.sum(0)
    dd = d1.double().reshape(-1, C)
    gh = ((g.double() - coef[2].double()) * coef[3].double()).reshape(-1, C)
    assert torch.allclose(s[0], dd.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[1], (dd * gh).sum(0), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('C,Wn,Wp', [(64, 8, 8), (128, 11, 16), (256, 16, 16)])
def test_bneck_wgrad_vs_torch(ops, dt, C, Wn, Wp):
    """Both weight gradients of the 1x1 convolutions (contraction over positions through two identity-MFMA transposes):
    dWe[o][n] = sum_p dz[p][o] yb[p][n], dbe = sum_p dz;  dWs[n][c] = sum_p dq[p][n] relu(bn1(g))[p][c], dbs = sum_p dq."""
    rows = 2 * 37 * 25 + 7
    wide = _randn(rows, C, seed=1, dt=dt)
    nrw = _randn(rows, Wp, seed=2, dt=dt)
    nrw[:, Wn:] = 0
    coef = torch.stack([_randn(C, seed=4).abs() + 0.5, _randn(C, seed=5)]).contiguous()
    dW, db = ops.bneck_wgrad(wide, nrw, True)
    ref = wide.double().t() @ nrw.double()
    assert dW.shape == (C, Wp) and db.shape == (C,)
    tol = 2e-5 * rows ** 0.5 * float(ref.abs().max() / rows ** 0.5 + 1.0)
    assert float((dW.double() - ref).abs().max()) <= tol, float((dW.double() - ref).abs().max())
    assert float((db.double() - wide.double().sum(0)).abs().max()) <= tol
    dW2, db2 = ops.bneck_wgrad(wide, nrw, False, pre=coef, pre_relu=True)
    u = torch.relu(wide.double() * coef[0].double() + coef[1].double()).to(dt).double()      # rounded like the forward operand
    ref2 = nrw.double().t() @ u
    assert dW2.shape == (Wp, C) and db2.shape == (Wp,)
    assert float((dW2.double() - ref2).abs().max()) <= tol, float((dW2.double() - ref2).abs().max())
    assert float((db2.double() - nrw.double().sum(0)).abs().max()) <= tol
    # additivity over row shards (every workgroup's slice and the reduce kernel)
    h = (rows // 2) // 32 * 32 + 5
    a, _ = ops.bneck_wgrad(wide[:h].contiguous(), nrw[:h].contiguous(), True)
    b, _ = ops.bneck_wgrad(wide[h:].contiguous(), nrw[h:].contiguous(), True)
    assert float((a + b - dW).abs().max()) <= tol


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('Wn,Wp,V,stride', [(8, 8, 25, 1), (11, 16, 25, 2), (16, 16, 18, 1), (8, 8, 18, 2)])
def test_bneck_wgrad_taps_vs_torch(ops, dt, Wn, Wp, V, stride):
    """Weight + bias gradient of the narrow 15-tap conv (frames transposed by 16x16x32 identity MFMAs as they enter the
    register ring) against autograd of F.conv2d in float64 on the same 16-bit tensors."""
    NM, T, k = 3, 43, 15
    Tz = (T - 1) // stride + 1
    q = _randn(NM, T, V, Wp, seed=1, dt=dt)
    dy = _randn(NM, Tz, V, Wp, seed=2, dt=dt)
    q[..., Wn:] = 0
    dy[..., Wn:] = 0
    taps, in_mul = ops.conv_taps_fwd(k, stride)
    dW, db = ops.bneck_wgrad_taps(dy, q, k, taps[0], in_mul=in_mul)
    w4 = torch.zeros(Wp, Wp, k, 1, dtype=torch.float64, device=dev(), requires_grad=True)
    out = F.conv2d(q.double().permute(0, 3, 1, 2), w4, None, (stride, 1), ((k - 1) // 2, 0))
    out.backward(dy.double().permute(0, 3, 1, 2))
    ref = w4.grad[..., 0].permute(2, 0, 1)                       # [k][n'][n]
    n = NM * Tz * V
    tol = 3e-5 * n ** 0.5 * (float(ref.abs().max()) / n ** 0.5 + 1.0)
    assert dW.shape == (k, Wp, Wp) and db.shape == (Wp,)
    assert float((dW.double() - ref).abs().max()) <= tol, float((dW.double() - ref).abs().max())
    assert float((db.double() - dy.double().sum((0, 1, 2))).abs().max()) <= tol
    if Wp > Wn:
        assert float(dW[:, Wn:].abs().max()) == 0.0 and float(dW[:, :, Wn:].abs().max()) == 0.0


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('C,Wn,Wp', [(64, 8, 8), (128, 11, 16)])
@pytest.mark.parametrize('p', [0.0, 0.5])
def test_bneck_bwd_in_equals_affine2_then_in_then_wgrad(ops, dt, C, Wn, Wp, p):
    """The fused backward input (BatchNorm-backward elementwise half + dropout mask formed in registers) against the three
    launches it replaces, on the same inputs and the same Philox seed: dz is rounded to the storage type in both, so dyb
    agrees to a rounding of the result and the weight / bias gradients to fp32 summation noise."""
    rows = 2 * 29 * 25 + 9
    dres = _randn(rows, C, seed=1, dt=dt)
    z = _randn(rows, C, seed=2, dt=dt)
    yb = _randn(rows, Wp, seed=3, dt=dt)
    yb[:, Wn:] = 0
    abc = torch.stack([_randn(C, seed=4).abs() + 0.3, _randn(C, seed=5) * 0.2, _randn(C, seed=6) * 0.05]).contiguous()
    We = _randn(C, Wn, seed=7, scale=C ** -0.5)
    seed = 1234567
    dz = ops.affine2(dres, z, abc, p, seed)
    dyb0 = ops.bneck_in(dz, We.t(), Wp)
    dW0, db0 = ops.bneck_wgrad(dz, yb, True)
    dyb1, dW1, db1 = ops.bneck_bwd_in(dres, z, abc, yb, We.t(), Wp, p_drop=p, seed=seed)
    _close(dyb1, dyb0, dt, 'dyb', extra=1.0)
    tol = 2e-5 * rows ** 0.5 * (float(dW0.abs().max()) / rows ** 0.5 + 1.0)
    assert float((dW1 - dW0).abs().max()) <= tol and float((db1 - db0).abs().max()) <= tol
    if p > 0:                                                  # the mask really is applied (and is the forward's)
        dz_nodrop = ops.affine2(dres, z, abc, 0.0, seed)
        assert float((dz.float() - dz_nodrop.float()).abs().max()) > 0.1


@pytest.mark.parametrize('dt', DT, ids=lambda d: str(d)[6:])
@pytest.mark.parametrize('tag,layout', [('st_gcn_mstcn_1x1', 'openpose'), ('st_gcn_mstcn_1x1_deep', 'ntu-rgb+d')])
def test_model_new_kernels_vs_generic_kernels(ops, dt, tag, layout):
    """Whole bottleneck models (all widths 8 / 11 -> 16 / 16, both stride-2 blocks), one training step with the
    register-chained bottleneck kernels and with the generic temporal-conv kernels in the same storage type, both against
    the SAME step of the float32 HIP path (pinned to the reference by the golden fixtures)."""
    import importlib
    from istgcn_amd import harness
    mod = importlib.import_module('istgcn_amd.net.' + tag)
    gargs = dict(layout=layout, strategy='spatial')
    Vj = 18 if layout == 'openpose' else 25
    res = {}
    try:
        for key, flag, d_ in (('f32', False, torch.float32), ('old', False, dt), ('new', True, dt)):
            ops.BNECK_RC = flag
            torch.manual_seed(0)
            m = mod.Model(3, 20, gargs, True, dropout=0, compute_dtype=d_)
            m.apply(harness.weights_init)
            m.to(dev()).train()
            gen = torch.Generator().manual_seed(11)
            x = torch.randn(4, 3, 64, Vj, 2, generator=gen).to(dev())
            y = torch.randint(0, 20, (4,), generator=gen).to(dev())
            ls = 1024.0 if d_ == torch.float16 else 1.0
            logits = m(x)
            loss = F.cross_entropy(logits, y)
            (loss * ls).backward()
            res[key] = (logits.detach().double().cpu(),
                        torch.cat([(p.grad.double() / ls).flatten().cpu() for p in m.parameters() if p.grad is not None]))
    finally:
        ops.BNECK_RC = True
    (l32, g32), (l0, g0), (l1, g1) = res['f32'], res['old'], res['new']
    assert torch.isfinite(g1).all()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    from gpu_util import gate16
    name = 'bneck_rc %s %s' % (tag, str(dt)[6:])
    # both 16-bit paths against the float32 HIP path (pinned to the reference by the golden tests): the new kernels must
    # not be noisier than the generic ones (small batch, 20+ ReLU masks: the gradient noise of 16-bit storage is large)
    el0, el1, eg0, eg1 = rel(l0, l32), rel(l1, l32), rel(g0, g32), rel(g1, g32)
    assert gate16(name + ' logits rel-L2 vs fp32 HIP (generic kernels: %.3g)' % el0, el1, 2e-2 if dt == torch.bfloat16 else 3e-3)
    assert gate16(name + ' whole-gradient rel-L2 vs fp32 HIP (generic kernels: %.3g)' % eg0, eg1, 0.5 if dt == torch.bfloat16 else 0.19)
    assert el1 < 1.5 * el0 + 1e-4 and eg1 < 1.3 * eg0 + 1e-3, (el0, el1, eg0, eg1)


def test_bneck_fullsize_config5_properties(ops):
    """Config-5 layer shape (128 clips x 2 persons -> NM = 256, T = 600, V = 25, 64 <-> 8 channels, float16): sequences are
    independent (a batch slice alone == the slice of the full launch, bit for bit: exercises the segment walk and the ring
    warm-up at every segment boundary), and the epilogue's BatchNorm sums equal the sums of the stored output."""
    dt = torch.float16
    NM, T, V, C, Wn, k = 256, 600, 25, 64, 8, 15
    g = _randn(NM, T, V, C, seed=1, dt=dt)
    Ws = _randn(Wn, C, seed=2, scale=C ** -0.5)
    coef = torch.stack([_randn(C, seed=3).abs() + 0.5, _randn(C, seed=4)]).contiguous()
    q = ops.bneck_in(g, Ws, 8, pre=coef, pre_relu=True)
    Wt = _randn(k, Wn, Wn, seed=5, scale=(k * Wn) ** -0.5)
    We = _randn(C, Wn, seed=6, scale=Wn ** -0.5)
    st = ops.new_stats(C, dev())
    yb, z = ops.bneck_out(q, Wt, list(range(k)), -7, We, C, stats=st, mode=0, Tout=T, Mlog=T)
    torch.cuda.synchronize()
    for lo, hi in ((0, 1), (101, 104), (NM - 2, NM)):
        qs = ops.bneck_in(g[lo:hi].contiguous(), Ws, 8, pre=coef, pre_relu=True)
        assert torch.equal(qs, q[lo:hi])
        ybs, zs = ops.bneck_out(qs, Wt, list(range(k)), -7, We, C, mode=0, Tout=T, Mlog=T)
        assert torch.equal(ybs, yb[lo:hi]) and torch.equal(zs, z[lo:hi]), 'sequence independence broken for [%d:%d)' % (lo, hi)
    s = st.sum(0)
    zd = z.double().reshape(-1, C)
    assert torch.allclose(s[0], zd.sum(0), rtol=1e-6, atol=1e-2)
    assert torch.allclose(s[1], (zd * zd).sum(0), rtol=1e-6, atol=1e-2)
    # time reversal: reversing the frames of q and the order of the taps reverses yb (exercises every tap slot and the
    # zero padding at both sequence ends at full depth)
    ybr, _ = ops.bneck_out(q.flip(1).contiguous(), Wt.flip(0).contiguous(), list(range(k)), -7, We, C, mode=0, Tout=T, Mlog=T)
    # (the taps are then accumulated in the opposite order: equal to one rounding of the storage type, not bit for bit)
    assert float((ybr.flip(1).float() - yb.float()).abs().max()) <= 2.0 ** -10 * float(yb.float().abs().max())
