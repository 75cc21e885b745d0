"""-m gpu: st_gcn blocks and whole Models of the HIP product (through the drop-in nn.Module API) against the golden
fixtures generated from the reference, in fp32 (tight) and bf16 storage (loose, stated)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from detinit import det_fill_, det_tensor, det_labels, WIDE_BLOCKS, wide_block_inputs, wide_block_has
from gpu_util import dev, diag, close, l2rel, sub_close

pytestmark = pytest.mark.gpu
SD = json.load(open(os.path.join(GOLDEN, 'state_dict_g5.json')))
MODEL_CFG = {
    'st_gcnold': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_msgcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_mstcn_1x1': (dict(layout='openpose', strategy='spatial'), 400),
    'st_gcn_multi3_fix_3A_mstcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_mstcn_1x1_deep': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_mstcn': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_msgcn_new': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_deep_msgcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
}


@pytest.fixture(scope='module')
def ops():
    from istgcn_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------ pointwise
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_bn_finalize_and_block_out(ops, dt):
    d = dev()
    g = torch.Generator().manual_seed(3)
    rows, C = 1000, 64
    z = torch.randn(rows, C, generator=g) * 2 + 0.5
    res = torch.randn(rows, C, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    if dt != torch.float32:
        z, res = z.to(dt).float(), res.to(dt).float()
    rm, rv = torch.zeros(C), torch.ones(C)
    ref = F.batch_norm(z, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    st = ops.new_stats(C, d)
    st[0, 0] = z.double().sum(0).to(d)
    st[3, 1] = (z.double() ** 2).sum(0).to(d)
    rmd, rvd = torch.zeros(C, device=d), torch.ones(C, device=d)
    coef = ops.bn_finalize(st, rows, gamma.to(d), beta.to(d), rmd, rvd, 0.1, 1e-5, True)
    assert rel_err(rmd, rm) < 1e-6 and rel_err(rvd, rv) < 1e-5
    zd = z.to(d, dt).view(10, 4, 25, C)
    out = ops.block_out_fwd(zd, coef[:2].contiguous(), res.to(d, dt).view_as(zd), None, 0.0, 0)
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert diag('block_out_fwd_%s' % str(dt)[6:], out.float().view(rows, C), F.relu(ref + res), tol) < tol
    # eval mode: running statistics
    coef_e = ops.bn_finalize(None, 0, gamma.to(d), beta.to(d), rmd, rvd, 0.1, 1e-5, False)
    ref_e = F.batch_norm(z, rmd.cpu(), rvd.cpu(), gamma, beta, training=False, eps=1e-5)
    out_e = ops.block_out_fwd(zd, coef_e[:2].contiguous(), None, None, 0.0, 0)
    assert diag('block_out_eval_%s' % str(dt)[6:], out_e.float().view(rows, C), F.relu(ref_e), tol) < tol


def test_dropout_mask_is_regenerated_in_backward(ops):
    d = dev()
    rows, C, p, seed = 4096, 64, 0.5, 1234567
    z = torch.randn(1, rows // 25 + 1, 25, C, device=d).abs() + 0.1
    coef = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(d)
    out = ops.block_out_fwd(z, coef[:2].contiguous(), None, None, p, seed)
    keep = out > 0
    frac = float(keep.float().mean())
    assert 0.47 < frac < 0.53
    assert rel_err(out[keep], (z * 2)[keep]) < 1e-6                     # 1/(1-p) scaling
    out2 = ops.block_out_fwd(z, coef[:2].contiguous(), None, None, p, seed)
    assert torch.equal(out, out2)
    out3 = ops.block_out_fwd(z, coef[:2].contiguous(), None, None, p, seed + 1)
    assert not torch.equal(out, out3)
    # backward path: affine2 with the same (p, seed) applies the same mask
    ones = torch.ones_like(z)
    abc = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C)]).to(d)
    dz = ops.affine2(ones, None, abc, p, seed)
    assert torch.equal(dz > 0, keep)
    dres, st2, _ = ops.block_out_bwd(ones, out, z, coef, None, None, p, seed)
    assert torch.equal(dres > 0, keep)
    assert rel_err(st2.sum(0)[0], (dres * dz).double().sum((0, 1, 2))) < 1e-6


# ------------------------------------------------------------------------------------------------ blocks (G3)
def _block_args(kind, A, A2, A3, imps, mst):
    if kind in ('st_gcnold',):
        return (A * imps[0],)
    if kind == 'st_gcn_msgcn':
        return (A * imps[0], A2 * imps[1], A3 * imps[2])
    if kind in ('st_gcn_mstcn', 'st_gcn_mstcn_1x1'):
        return (A * imps[0], mst)
    return (A, imps[0], imps[1], imps[2], mst)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('kind', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn'])
def test_blocks_golden(golden, kind, dt):
    g = golden('block_g3_%s.npz' % kind)
    mod = importlib.import_module('istgcn_amd.net.' + kind)
    d = dev()
    bases = sorted({'.'.join(k.split('.')[:2]) + '.' for k in g.files})
    # 16-bit storage: relative L2 (ReLU-mask flips on 576-position fixtures); float16 has 3 more mantissa bits than bfloat16
    # (gradients: one ReLU-mask flip on these 576-position fixtures moves a BatchNorm-bias gradient entry by O(1) whatever
    #  the 16-bit format -- measured 0.11 in float16 -- so both formats share the gradient gate; outputs are 4x tighter)
    # measured round 3 (max over all fixtures, gpurun_out/err16_measured.txt): bf16 outputs .0062, gradients .149; fp16 outputs
    # 8.3e-4, gradients .058 -> gates at <= 2x
    # (the bf16 gradient gate was 0.15 against a largest entry of 0.1486 -- tcn_end.0.bias of the full IST-GCN block, the same
    #  value in five runs on the final build: deterministic, but 1 % from the gate; 0.2 keeps it a gate on the entries at
    #  0.08-0.13 without hanging the suite on the last digit of one ReLU-mask flip)
    tol_f, tol_g = {torch.float32: (3e-5, 2e-4), torch.bfloat16: (1.25e-2, 0.2), torch.float16: (1.7e-3, 0.12)}[dt]
    if dt != torch.float32 and kind == 'st_gcn_mstcn_1x1':
        # the fixture's bottleneck is int(sqrt(16)) = 4 channels wide: 16-bit storage of a 4-channel tensor is mostly noise
        # (measured .259 / .045)
        tol_g = 0.4 if dt == torch.bfloat16 else 0.09

    for b in bases:
        t = lambda k: torch.from_numpy(g[b + k])  # noqa: E731
        sd = {k[len(b) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(b + 'sd.')}
        A, A2, A3, x, r = t('A').to(d), t('A2').to(d), t('A3').to(d), t('x'), t('r')
        wkey = 'gcn.conv.weight' if 'gcn.conv.weight' in sd else 'gcn.branch.conv.weight'
        K = A.shape[0]
        cout, cin = sd[wkey].shape[0] // K, sd[wkey].shape[1]
        stride = x.shape[2] // r.shape[2]
        blk = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=(b.split('.')[1] != 's0'))
        blk.load_state_dict(sd, strict=True)
        blk.to(d)
        imps = [t('imp%d' % j).to(d).requires_grad_(True) for j in (1, 2, 3)]
        mst = t('mst').to(d).requires_grad_(True)
        xin = x.to(d, dt)
        name = 'blk_%s_%s_%s' % (kind, b.strip('.').replace('.', '_'), str(dt)[6:])
        blk.eval()
        with torch.no_grad():
            y = blk(xin, *_block_args(kind, A, A2, A3, imps, mst))[0]
        assert close(name + '_yeval', y.float(), g[b + 'y_eval'], tol_f, dt)
        blk.train()
        xx = xin.clone().requires_grad_(True)
        y = blk(xx, *_block_args(kind, A, A2, A3, imps, mst))[0]
        assert close(name + '_ytrain', y.float(), g[b + 'y_train'], tol_f, dt)
        (y.float() * r.to(d)).sum().backward()
        assert close(name + '_dx', xx.grad.float(), g[b + 'dx'], tol_g, dt)
        n_grad = 0
        for k, p in blk.named_parameters():
            if b + 'grad.' + k in g.files:
                assert p.grad is not None, k
                assert close(name + '_grad_' + k, p.grad, g[b + 'grad.' + k], tol_g, dt), k
                n_grad += 1
            else:
                assert p.grad is None, k
        assert n_grad > 0
        for j in (1, 2, 3):
            if b + 'dimp%d' % j in g.files:
                assert close(name + '_dimp%d' % j, imps[j - 1].grad, g[b + 'dimp%d' % j], tol_g, dt)
        if b + 'dmst' in g.files and not (dt != torch.float32 and kind == 'st_gcn_mstcn_1x1'):
            # (the bf16 bottleneck fixture has width int(sqrt(16)) = 4: its 3 importance gradients are sums of ~200
            #  strongly cancelling products of bf16-rounded terms; checked in fp32 only)
            assert close(name + '_dmst', mst.grad, g[b + 'dmst'], tol_g, dt)
        for k, v in blk.state_dict().items():
            if 'running' in k:
                assert close(name + '_' + k, v, g[b + 'after.' + k], tol_f, dt), k


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('kind', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn'])
def test_blocks_wide_golden(golden, kind, dt):
    """every block variant at the trunk's widths -- 64 -> 64 (stride 1), 64 -> 128 (stride 2; bottleneck
    widths 8 and 11), 128 -> 256 (stride 2, 1 x 1 residual conv) and 256 -> 256 (bottleneck width 16), V = 25, and (config 3's
    bottleneck block, the plain block) V = 18 -- against REFERENCE-generated outputs,
    input / parameter / importance gradients and running statistics (block_g3w_*.npz from net/st_gcnold.py:197-203,
    st_gcn_msgcn.py:231-237, st_gcn_mstcn.py:235-249, st_gcn_mstcn_1x1.py:250-266, st_gcn_multi3_fix_3A_mstcn.py:206-220).
    These shapes take the register-chained graph-conv kernels, the lean temporal conv and (16-bit `1x1`) the bottleneck
    stream kernels.  800 positions x 64-128 channels average the ReLU-mask flips out: 16-bit gradient gates of 0.1 / 0.06
    for every variant (the 16-channel fixtures of test_blocks_golden need 0.2 / 0.4)."""
    from istgcn_amd import ops, functional
    from istgcn_amd.net.utils.graph import Graph
    g = golden('block_g3w_%s.npz' % kind)
    mod = importlib.import_module('istgcn_amd.net.' + kind)
    d = dev()
    tol_f, tol_g = {torch.float32: (3e-5, 2e-4), torch.bfloat16: (1.25e-2, 0.1), torch.float16: (1.7e-3, 0.06)}[dt]
    seen = 0
    for si, (cin, cout, stride, V) in enumerate(WIDE_BLOCKS):
        if not wide_block_has(kind, si):
            continue
        seen += 1
        b = 'w%d.' % si
        gr = Graph('ntu-rgb+d' if V == 25 else 'openpose', 'spatial_3')
        A, A2, A3 = (torch.tensor(a, dtype=torch.float32, device=d) for a in (gr.A, gr.A2, gr.A3))
        K = A.shape[0]
        x, r = wide_block_inputs(si, kind=kind)
        blk = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=True)
        blk.load_state_dict(det_fill_(blk.state_dict(), salt=100 + si), strict=True)
        blk.to(d)
        if kind == 'st_gcn_mstcn_1x1' and dt != torch.float32:
            w = int(cout ** 0.5)
            assert ops.bneck_ok(V, cout, w, functional._pad_width(w, dt), dt), 'the bottleneck stream kernels serve this block'
        imps = [torch.from_numpy(g[b + 'imp%d' % j]).to(d).requires_grad_(True) for j in (1, 2, 3)]
        mst = torch.from_numpy(g[b + 'mst']).to(d).requires_grad_(True)
        xin = x.to(d, dt)
        name = 'blkw_%s_w%d_%s' % (kind, si, str(dt)[6:])
        blk.eval()
        with torch.no_grad():
            y = blk(xin, *_block_args(kind, A, A2, A3, imps, mst))[0]
        assert sub_close(name + '_yeval', y.float(), g, b + 'y_eval', tol_f, dt)
        blk.train()
        xx = xin.clone().requires_grad_(True)
        with ops.trace() as tr:
            y = blk(xx, *_block_args(kind, A, A2, A3, imps, mst))[0]
            assert sub_close(name + '_ytrain', y.float(), g, b + 'y_train', tol_f, dt)
            (y.float() * r.to(d)).sum().backward()
        if dt != torch.float32:
            # the kernels this fixture is meant to pin are the ones that ran (the library's dispatch trace, not its predicates):
            # register-chained graph conv forward / weight gradient / data gradient WITH the adjacency gradient (256 output
            # channels -- cases 3 / 4 -- included since round 5); the lean temporal conv and its lean weight gradient, or the
            # bottleneck stream kernels
            want = ['gcn_rc_fwd_kernel', 'gcn_rc_wgrad_kernel', 'gcn_rc_bwd_kernel']
            if kind == 'st_gcn_mstcn_1x1':
                want += ['bneck_']
            elif stride == 2 and kind in ('st_gcn_mstcn', 'st_gcn_multi3_fix_3A_mstcn'):
                # the 15-tap fold at stride 2: a 33-frame window (forward) and 8 + 7 taps per output phase (data gradient) are
                # outside tconv_lean_geom() -> the round-3 kernel; the weight gradient is lean (8 + 7 taps per parity)
                want += ['tconv_kernel', 'twg_lean_kernel']
            else:
                want += ['tconv_lean_kernel', 'twg_lean_kernel']
            # (the three dispatch overrides the library keeps select the previous generation of a family; this test runs under
            #  each of them in tests/test_gpu_overrides.py, and then THOSE kernels must be the ones that ran)
            older = {}
            if os.environ.get('ISTGCN_GCN_RC') == '0':
                older.update(gcn_rc_fwd_kernel='gcn_fwd_kernel', gcn_rc_wgrad_kernel='tconv_wgrad_kernel', gcn_rc_bwd_kernel='gcn_bwd_kernel')
            if os.environ.get('ISTGCN_TCONV_LEAN') == '0':
                older.update(tconv_lean_kernel='tconv_kernel')
            if os.environ.get('ISTGCN_TWG_LEAN') == '0':
                older.update(twg_lean_kernel='tconv_wgrad_kernel')
            assert not [w for w in older if w in want and tr.ran(w)], ('override ignored', sorted(tr.kernels))
            want = [older.get(w, w) for w in want]
            missing = [w for w in want if not tr.ran(w)]
            assert not missing, (missing, sorted(tr.kernels))
        assert sub_close(name + '_dx', xx.grad.float(), g, b + 'dx', tol_g, dt)
        n_grad = 0
        for k, p in blk.named_parameters():
            if b + 'grad.' + k in g.files:
                assert p.grad is not None, k
                # (the 64 / 128-element bias and BatchNorm-shift gradients are sums with heavy cancellation: in bf16 storage
                #  their rel-L2 error is 0.07-0.10 over the ten fixtures, gpurun_out/err16_measured.txt -- 1.3 x the gate of the
                #  weight tensors; fp32 and fp16 keep one gate)
                tg = 1.3 * tol_g if (dt == torch.bfloat16 and k.endswith('bias')) else tol_g
                # (a structurally-zero gradient -- a conv bias in front of a batch-statistics BatchNorm -- is the rounding noise of
                #  a sum over positions x channels of 16-bit-rounded terms: it grows with the square root of the channel count)
                assert sub_close(name + '_grad_' + k, p.grad, g, b + 'grad.' + k, tg, dt, zero_gate=0.5 * (cout / 64.0) ** 0.5), k
                n_grad += 1
            else:
                assert p.grad is None, k
        assert n_grad > 0
        for j in (1, 2, 3):
            if b + 'dimp%d' % j in g.files:
                assert sub_close(name + '_dimp%d' % j, imps[j - 1].grad, g, b + 'dimp%d' % j, tol_g, dt)
        if b + 'dmst' in g.files:
            # (three numbers, each a sum over all 10^5 output elements of strongly cancelling products of 16-bit-rounded
            #  terms -- the branch outputs are recomputed from the folded taps: measured 0.106 in bfloat16, 1.5x the gate)
            # (the bottleneck block's branches are 8 / 11 channels wide: measured 0.22 in bfloat16, 0.034 in float16 -- the
            #  ratio of the two formats' rounding steps, i.e. rounding noise, not a defect)
            # (and 0.20 in float16 on the round-1/2 graph-conv kernels -- ISTGCN_GCN_RC=0, tests/test_gpu_overrides.py --, which
            #  round the aggregated tile to 16 bits in LDS before the channel contraction)
            gate = 0.3 if (kind == 'st_gcn_mstcn_1x1' and (dt == torch.bfloat16 or os.environ.get('ISTGCN_GCN_RC') == '0')) else 1.5 * tol_g
            assert close(name + '_dmst', mst.grad, g[b + 'dmst'], gate, dt)
        for k, v in blk.state_dict().items():
            if 'running' in k:
                assert close(name + '_' + k, v, g[b + 'after.' + k], tol_f, dt), k
    assert seen >= 2


# ------------------------------------------------------------------------------------------------ models (G4/G5)
def _model(tag, dt, dropout=0):
    gargs, nc = MODEL_CFG[tag]
    mod = importlib.import_module('istgcn_amd.net.' + tag)
    m = mod.Model(3, nc, gargs, True, dropout=dropout, compute_dtype=dt)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == SD[tag]
    m.load_state_dict(det_fill_(m.state_dict()))
    return m.to(dev()), nc


@pytest.mark.parametrize('tag', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn', 'st_gcn_mstcn_1x1_deep'])
def test_model_eval_logits_full_clip(golden, tag):
    """BASELINE.json:north_star parity bar: logits within 1e-3 (fp32) of the reference forward on identical
    (N,C,T,V,M) inputs, at the full clip shape of each config (N=2)."""
    g = golden('model_g4_%s.npz' % tag)
    m, nc = _model(tag, torch.float32)
    m.eval()
    x = det_tensor('g4.x.' + tag, tuple(int(s) for s in g['eval_shape'])).to(dev())
    with torch.no_grad():
        y = m(x)
    assert diag('model_eval_' + tag, y, g['eval_logits'], 1e-3) < 1e-3


@pytest.mark.parametrize('tag', sorted(MODEL_CFG))
def test_model_train_step_fp32(golden, tag):
    """one SGD-nesterov step = processor/recognition.py:249-296: logits, loss, every gradient norm, every updated
    parameter norm and a few updated tensors against the reference."""
    from istgcn_amd import harness
    g = golden('model_g4_%s.npz' % tag)
    m, nc = _model(tag, torch.float32)
    shp = tuple(int(s) for s in g['train_shape'])
    x = det_tensor('g4.xt.' + tag, shp).to(dev())
    lab = det_labels('g4.lab.' + tag, shp[0], nc).to(dev())
    opt = harness.make_optimizer(m)
    m.train()
    logits = m(x)
    loss = F.cross_entropy(logits, lab)
    opt.zero_grad()
    loss.backward()
    assert diag('model_train_logits_' + tag, logits, g['train_logits'], 1e-3) < 1e-3
    assert abs(float(loss.detach()) - float(g['train_loss'])) < 1e-3
    params = list(m.parameters())
    none = np.asarray([p.grad is None for p in params])
    assert np.array_equal(none, g['grad_none'])                     # dead params stay grad-less
    gn = np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) for p in params])
    bad = np.abs(gn - g['grad_norms']) > 2e-3 * np.maximum(1.0, g['grad_norms'])
    assert not bad.any(), [(SD[tag + '#param_names'][i], gn[i], g['grad_norms'][i]) for i in np.flatnonzero(bad)[:8]]
    opt.step()
    sd = m.state_dict()
    after = np.asarray([float(sd[k].double().norm()) for k in SD[tag + '#param_names']])
    assert np.allclose(after, g['param_norms_after'], rtol=1e-3, atol=1e-5)
    bufn = np.asarray([float(v.double().norm()) for k, v in sd.items() if 'running' in k])
    assert np.allclose(bufn, g['buffer_norms_after'], rtol=1e-3, atol=1e-5)
    for k in g.files:
        if k.startswith('after.'):
            assert diag('model_after_%s_%s' % (tag, k), sd[k[6:]], g[k], 1e-3) < 1e-3, k


@pytest.mark.parametrize('tag', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn', 'st_gcn_mstcn_1x1_deep'])
def test_model_full_clip_train_step_reference(golden, tag):
    """G4L (round 5): one training step of every BASELINE model at its FULL clip length (8 clips x T = 300;
    4 x T = 600 for the deep model) against the REFERENCE -- processor/recognition.py:273-283 on net/<tag>.py: logits and loss
    to 1e-3, every parameter's gradient norm to 2e-3, a subsample of five gradient tensors element-wise, in fp32; the logits in
    both 16-bit storage types.  16 (8) sequences x T x V positions per layer: every kernel walks several tiles per workgroup,
    which the (4, T = 48) fixture of test_model_train_step_fp32 does not reach."""
    g = golden('model_g4l_%s.npz' % tag)
    shp = tuple(int(v) for v in g['train_shape'])
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        m, nc = _model(tag, dt)
        x = det_tensor('g4l.x.' + tag, shp).to(dev())
        lab = det_labels('g4l.lab.' + tag, shp[0], nc).to(dev())
        m.train()
        logits = m(x)
        loss = F.cross_entropy(logits.float(), lab)
        if dt != torch.float32:
            from gpu_util import gate16
            assert gate16('g4l_logits_%s %s' % (tag, str(dt)[6:]), l2rel(logits.float(), g['train_logits']), 3e-3 if dt == torch.bfloat16 else 4e-4)      # (measured 1.0-1.1e-3 / 1.1-1.5e-4)
            continue
        loss.backward()
        assert diag('g4l_logits_' + tag, logits, g['train_logits'], 1e-3) < 1e-3
        assert abs(float(loss.detach()) - float(g['train_loss'])) < 1e-3
        params = list(m.parameters())
        assert np.array_equal(np.asarray([p.grad is None for p in params]), g['grad_none'])
        gn = np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) for p in params])
        bad = np.abs(gn - g['grad_norms']) > 2e-3 * np.maximum(1.0, g['grad_norms'])
        assert not bad.any(), [(SD[tag + '#param_names'][i], gn[i], g['grad_norms'][i]) for i in np.flatnonzero(bad)[:8]]
        named = dict(m.named_parameters())
        n_sub = 0
        for k in g.files:
            if k.startswith('grad.') and not k.endswith('#norm'):
                assert sub_close('g4l_%s_%s' % (tag, k), named[k[5:]].grad, g, k, 2e-3, dt), k
                n_sub += 1
        assert n_sub >= 3


@pytest.mark.parametrize('tag', ['st_gcn_msgcn', 'st_gcn_multi3_fix_3A_mstcn'])
def test_bench_shape_train_step_reference(golden, tag):
    """G4B (round 5): the training steps bench.py times -- `net/st_gcn_msgcn.py` (BASELINE config 2, the headline) and the full
    IST-GCN `net/st_gcn_multi3_fix_3A_mstcn.py` (config 4), 64 clips x (3, 300, 25, 2) -- against the REFERENCE at exactly that
    shape (processor/recognition.py:273-283; fixtures generated by running the reference on the CPU: two minutes and 25 / 45 GB
    each).  fp32 storage: logits and loss 1e-3, every gradient norm 2e-3, subsampled gradient tensors 2e-3.  bf16 (the bench's
    storage type) and fp16: logits in relative L2, and the whole gradient's norm-weighted error against the reference's
    gradient norms."""
    g = golden('model_g4b_%s.npz' % tag)
    shp = tuple(int(v) for v in g['train_shape'])
    assert shp == (64, 3, 300, 25, 2)
    from gpu_util import gate16
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        m, nc = _model(tag, dt)
        x = det_tensor('g4b.x.' + tag, shp).to(dev())
        lab = det_labels('g4b.lab.' + tag, shp[0], nc).to(dev())
        m.train()
        logits = m(x)
        loss = F.cross_entropy(logits.float(), lab)
        # (float16 storage trains with the static loss scale of bench.py -- 65536 -- or the activation gradients of a 64-clip
        #  mean loss underflow: measured 0.16 norm error without it, recognition.py's --half has no scale at all)
        ls = 65536.0 if dt == torch.float16 else 1.0
        (loss * ls).backward()
        params = list(m.parameters())
        gn = np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) / ls for p in params])
        if dt == torch.float32:
            assert diag('g4b_logits_' + tag, logits, g['train_logits'], 1e-3) < 1e-3
            assert abs(float(loss.detach()) - float(g['train_loss'])) < 1e-3
            assert np.array_equal(np.asarray([p.grad is None for p in params]), g['grad_none'])
            bad = np.abs(gn - g['grad_norms']) > 2e-3 * np.maximum(1.0, g['grad_norms'])
            assert not bad.any(), [(SD[tag + '#param_names'][i], gn[i], g['grad_norms'][i]) for i in np.flatnonzero(bad)[:8]]
            named = dict(m.named_parameters())
            for k in g.files:
                if k.startswith('grad.') and not k.endswith('#norm'):
                    assert sub_close('g4b_%s_%s' % (tag, k), named[k[5:]].grad, g, k, 2e-3, dt), k
        else:
            name = tag + ' ' + str(dt)[6:]
            assert gate16('g4b_logits ' + name, l2rel(logits.float(), g['train_logits']), 3e-3 if dt == torch.bfloat16 else 4e-4)
            # every parameter's gradient NORM against the reference's (the norms of the reference are all the fixture holds of
            # most tensors): relative error of the norm, weighted by the norm -- a 16-bit regression alarm, not a parity gate
            w = g['grad_norms']
            e = float(np.abs(gn - w).sum() / w.sum())
            assert gate16('g4b_grad_norms ' + name, e, 1.2e-2 if dt == torch.bfloat16 else 4e-3)      # (measured 3.6-3.9e-3 / 1.3e-3)
        del m, logits, loss
        torch.cuda.empty_cache()


@pytest.mark.parametrize('tag', ['st_gcn_msgcn', 'st_gcn_multi3_fix_3A_mstcn', 'st_gcn_mstcn_1x1'])
def test_model_bf16_storage_close_to_fp32(golden, tag):
    """bf16 activations with fp32 accumulation / fp64 statistics against (a) the reference's fp32 logits and (b) the
    fp32 HIP path's gradients (itself pinned to the reference above): logits within 3e-2 relative L2, the full
    gradient within 0.25 relative L2, every sizeable parameter gradient with cosine > 0.8 (the deterministic test init
    is deliberately ill-conditioned -- eval logits up to 98 -- so bf16 ReLU-mask flips show; fp32 is the parity gate)."""
    g = golden('model_g4_%s.npz' % tag)
    shp = tuple(int(s) for s in g['train_shape'])
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        m, nc = _model(tag, dt)
        x = det_tensor('g4.xt.' + tag, shp).to(dev())
        lab = det_labels('g4.lab.' + tag, shp[0], nc).to(dev())
        m.train()
        logits = m(x)
        loss = F.cross_entropy(logits, lab)
        loss.backward()
        grads[dt] = ([None if p.grad is None else p.grad.double().flatten() for p in m.parameters()], logits.detach(), float(loss))
    from gpu_util import gate16
    g32, g16 = grads[torch.float32], grads[torch.bfloat16]
    gl, gg, gc = BF16_MODEL_GATE.get(tag, (3e-2, 0.25, 0.2))
    assert gate16('model_bf16 %s train logits rel-L2 vs reference' % tag, l2rel(g16[1], g['train_logits']), gl)
    assert abs(g16[2] - float(g['train_loss'])) < 5e-2
    a = torch.cat([t for t in g32[0] if t is not None])
    b = torch.cat([t for t in g16[0] if t is not None])
    assert torch.isfinite(b).all()
    assert gate16('model_bf16 %s whole-gradient rel-L2 vs fp32 HIP' % tag, float((a - b).norm() / a.norm()), gg)
    top = max(float(t.norm()) for t in g32[0] if t is not None)
    worst, wsum, wnorm = 0.0, 0.0, 0.0
    for ta, tb in zip(g32[0], g16[0]):
        if ta is not None and float(ta.norm()) > 1e-2 * top:
            d = 1.0 - float((ta * tb).sum() / (ta.norm() * tb.norm()))
            worst = max(worst, d)
            wsum += float(ta.norm()) * d
            wnorm += float(ta.norm())
    # the worst tensor is a maximum over ~100 tensors of a quantity that moves with the order of the atomics (which ReLU
    # masks flip): over 12 runs in one process it ranged 0.087-0.164 for st_gcn_mstcn_1x1 (and once left 0.2 in a suite
    # run), so its gate is a coarse one; the norm-weighted mean of the same quantity is the stable statistic
    assert gate16('model_bf16 %s worst (1 - cosine) over sizeable parameter gradients' % tag, worst, gc)
    assert gate16('model_bf16 %s norm-weighted mean (1 - cosine) over sizeable parameter gradients' % tag, wsum / wnorm,
                  BF16_MODEL_COS_MEAN_GATE[tag])


def test_extract_feature_shapes():
    m, nc = _model('st_gcnold', torch.float32)
    m.eval()
    x = torch.randn(2, 3, 32, 25, 2, device=dev())
    with torch.no_grad():
        out, feat = m.extract_feature(x)
        y = m(x)
    assert out.shape == (2, nc, 8, 25, 2) and feat.shape == (2, 256, 8, 25, 2)
    # pooling the per-position scores reproduces forward (fcn is linear)
    assert rel_err(out.mean(dim=(2, 3, 4)), y) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_native_packers_match_specification(ops, dt):
    """istgcn_pack_* (one launch, strided in-place reads) == the torch-op specification, bit for bit."""
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    for cin, cout, K in ((3, 64, 3), (64, 64, 3), (64, 128, 3), (256, 256, 3), (40, 24, 2), (64, 64, 4), (16, 8, 1)):
        w = torch.randn(K * cout, cin, 1, 1, generator=g).to(d)                       # Conv2d(Cin, K*Cout, 1).weight
        W3 = w.view(K, cout, cin)
        ref = ops.pack_gcn_weight_ref(W3.permute(1, 0, 2).contiguous().cpu(), dt)
        got = ops.pack_gcn_weight(W3.permute(1, 0, 2), dt)                            # permuted view, read in place
        assert torch.equal(got.cpu().view(ref.shape), ref)
        refb = ops.pack_gcn_wb_ref(W3.cpu(), dt)
        gotb = ops.pack_gcn_wb(W3, dt)
        assert torch.equal(gotb.cpu().view(refb.shape), refb)
    V = 25
    for cin, cout, k, s in ((64, 64, 9, 1), (64, 128, 9, 2), (256, 256, 9, 1), (8, 8, 15, 1), (3, 64, 1, 2), (24, 40, 3, 1)):
        w = torch.randn(cout, cin, k, 1, generator=g).to(d)                           # Conv2d(C, C, (k,1)).weight
        wf = w.view(cout, cin, k).permute(2, 0, 1)                                    # [k][Cout][Cin] view
        taps, in_mul = ops.conv_taps_fwd(k, s)
        ref = ops.pack_tconv_weight_ref(wf.contiguous().cpu(), V, taps, in_mul, dt)
        got = ops.pack_tconv_weight(wf, V, taps, in_mul, dt)
        assert torch.equal(got.cpu().view(ref.shape), ref)
        for phase in range(s):                                                        # data-gradient packs
            tl = ops.conv_taps_bwd(k, s, phase)
            if not tl:
                continue
            sel, offs = [j for j, _ in tl], [dj for _, dj in tl]
            wt = torch.stack([wf[j].t() for j in sel]).contiguous().cpu()
            ref = ops.pack_tconv_weight_ref(wt, V, offs, 1, dt)
            got = ops.pack_tconv_weight(wf.transpose(1, 2), V, offs, 1, dt, tap_sel=sel)
            assert torch.equal(got.cpu().view(ref.shape), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['plain', 'incep', '3a'])
def test_fused_importance_fold_matches_torch_spec(kind):
    """FoldFn (one launch each way) == fold_adjacency + fold_bias_term and their autograd."""
    from istgcn_amd import functional as Fn
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    K, V, C = 3, 25, 64
    A = (torch.rand(K, V, V, generator=g) < 0.15).float() * torch.rand(K, V, V, generator=g)
    A2, A3 = A.roll(1, 1) * 0.5, A.roll(2, 2) * 0.25
    mats = {'plain': [A], 'incep': [A, A2, A3], '3a': [A, A * A, A * A * A]}[kind]
    B = torch.stack(mats).to(d).contiguous()
    imps = [torch.rand(K, V, V, generator=g).to(d).requires_grad_() for _ in mats]
    bias = torch.randn(K * C, generator=g).to(d).requires_grad_()
    dA = torch.randn(K, V, V, generator=g).to(d)
    dS = torch.randn(V, C, generator=g).to(d)
    A_eff, bterm = Fn.FoldFn.apply(B, bias, C, *imps)
    (A_eff * dA).sum().add((bterm * dS).sum()).backward()
    got = [A_eff.detach(), bterm.detach(), bias.grad.clone()] + [i.grad.clone() for i in imps]
    for t in imps + [bias]:
        t.grad = None
    ref_A = Fn.fold_adjacency(kind, A.to(d), imps, A2.to(d), A3.to(d))
    ref_b = Fn.fold_bias_term(bias, ref_A, C)
    (ref_A * dA).sum().add((ref_b * dS).sum()).backward()
    ref = [ref_A.detach(), ref_b.detach(), bias.grad] + [i.grad for i in imps]
    for a, b in zip(got, ref):
        assert rel_err(a.cpu(), b.cpu()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('co,ci,scale', [(64, 64, 1.0), (256, 256, 1.0 / 3.0), (11, 11, 1.0), (8, 16, 0.5)])
def test_fused_tcn_tap_fold_matches_torch_spec(co, ci, scale):
    """TcnTapsFn (one launch each way) == fold_tcn_taps and its autograd, all seven parameter gradients."""
    from istgcn_amd import functional as Fn
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(4)
    ws = [torch.randn(co, ci, k, 1, generator=g).to(d).requires_grad_() for k in (3, 9, 15)]
    bs = [torch.randn(co, generator=g).to(d).requires_grad_() for _ in range(3)]
    mst = (0.5 + torch.rand(3, generator=g)).to(d).requires_grad_()
    dT = torch.randn(15, co, ci, generator=g).to(d)
    dB = torch.randn(co, generator=g).to(d)
    params = ws + bs + [mst]
    taps, bias = Fn.TcnTapsFn.apply(*ws, *bs, mst, scale)
    (taps * dT).sum().add((bias * dB).sum()).backward()
    got = [taps.detach(), bias.detach()] + [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    rt, rb = Fn.fold_tcn_taps(*ws, *bs, mst, scale)
    (rt * dT).sum().add((rb * dB).sum()).backward()
    ref = [rt.detach(), rb.detach()] + [p.grad for p in params]
    for a, b in zip(got, ref):
        assert a.shape == b.shape and rel_err(a.cpu(), b.cpu()) < 5e-6


# ------------------------------------------------------------------------------------------------ round-2 additions
def test_flat_sgd_kernel_matches_torch_sgd():
    """harness.FlatSGD (one istgcn_sgd_step launch over the flat buffers) against torch.optim.SGD (recognition.py:
    154-159): three steps, a dead parameter (no gradient -> untouched, no weight decay), a late one (slow path)."""
    from istgcn_amd import harness
    d = dev()
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 1, 1), (64,), (7, 5), (1,), (3, 25, 25), (33,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(d)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(d)) for t in init]
    opt_a = harness.FlatSGD(pa, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    opt_b = torch.optim.SGD(pb, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    dead, late = 3, 5
    for step in range(4):
        opt_a.zero_grad()
        opt_b.zero_grad()
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == dead or (i == late and step < 2):
                continue
            gr = torch.randn(a.shape, generator=g).to(d)
            a.grad = gr.clone()
            b.grad = gr.clone()
        if step == 2:
            for grp in list(opt_a.param_groups) + list(opt_b.param_groups):
                grp['lr'] = 0.01                                    # recognition.py:168-176 adjust_lr
        opt_a.step()
        opt_b.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert rel_err(a, b) < 2e-6, (step, i)
    assert torch.equal(pa[dead].detach().cpu(), init[dead])           # never touched
    assert opt_a.bucket_bytes == 4 * sum((p.numel() + 3) // 4 * 4 for i, p in enumerate(pa) if i not in (dead, late))
    # loss scale: gradients scaled by S and un-scaled inside the kernel give the same update
    pc = [torch.nn.Parameter(t.clone().to(d)) for t in init[:2]]
    pd = [torch.nn.Parameter(t.clone().to(d)) for t in init[:2]]
    oc, od = harness.FlatSGD(pc, loss_scale=1024.0), harness.FlatSGD(pd)
    for c, dd in zip(pc, pd):
        gr = torch.randn(c.shape, generator=g).to(d)
        c.grad, dd.grad = gr * 1024.0, gr.clone()
    oc.step()
    od.step()
    for c, dd in zip(pc, pd):
        assert rel_err(c, dd) < 1e-6


def test_flat_sgd_skips_non_finite_gradients_and_resumes_exactly():
    """(a) with a loss scale in use an inf / NaN anywhere in the gradients skips the WHOLE update
    (GradScaler's rule: no partial step with an overflowed scale) and raises the flag check_overflow() polls, which backs
    the loss scale off; the step after it is applied normally; with skip_nonfinite=False only the non-finite elements
    are left out; (b) load_state_dict on a fresh optimizer puts the momentum in place before the first update: a resumed
    run continues bit for bit."""
    from istgcn_amd import harness
    d = dev()
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(33, generator=g), torch.randn(4, 5, generator=g)]
    grads = [[torch.randn(t.shape, generator=g) for t in init] for _ in range(4)]

    def poisoned(skip):
        ps = [torch.nn.Parameter(t.clone().to(d)) for t in init]
        o = harness.FlatSGD(ps, lr=0.1, loss_scale=1024.0, skip_nonfinite=skip, poll_every=0)
        for p, gr in zip(ps, grads[0]):
            p.grad = (gr * 1024.0).to(d)
        o.step()                                           # a clean step fixes the layout (and gives the momentum a value)
        before = [p.detach().clone() for p in ps]
        mom = o.M.clone()
        for p, gr in zip(ps, grads[1]):
            p.grad = (gr * 1024.0).to(d)
        ps[0].grad[5] = float('inf')
        ps[1].grad[2, 3] = float('nan')
        o.step()
        return ps, o, before, mom

    pa, oa, before, mom = poisoned(None)                   # default with a loss scale: skip the whole step
    assert oa.skip_nonfinite
    assert all(torch.equal(p.detach(), b) for p, b in zip(pa, before)) and torch.equal(oa.M, mom)
    assert oa.check_overflow() and oa.loss_scale == 512.0 and not oa.check_overflow()
    for p, gr in zip(pa, grads[2]):                        # the next (finite) step is applied
        p.grad = (gr * 512.0).to(d)
    oa.step()
    assert all(not torch.equal(p.detach(), b) for p, b in zip(pa, before)) and not oa.check_overflow()
    pe, oe, before, mom = poisoned(False)                  # element-wise: only the non-finite elements are left out
    assert torch.isfinite(oe.P).all() and torch.isfinite(oe.M).all()
    assert float(pe[0][5]) == float(before[0][5]) and float(pe[1][2, 3]) == float(before[1][2, 3])
    assert float(pe[0][6]) != float(before[0][6])
    assert oe.check_overflow() and oe.loss_scale == 512.0 and not oe.check_overflow()
    # exact resume: two steps, save, two more  ==  two steps, save | fresh optimizer + load, two more
    def run(resume):
        ps = [torch.nn.Parameter(t.clone().to(d)) for t in init]
        o = harness.FlatSGD(ps, lr=0.1)
        for s in range(4):
            if resume and s == 2:
                sd = o.state_dict()
                ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
                o = harness.FlatSGD(ps, lr=0.1)
                o.load_state_dict(sd)
            for p, gr in zip(ps, grads[s]):
                p.grad = gr.clone().to(d)
            o.step()
        return [p.detach().clone() for p in ps]
    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)
    # a parameter re-pointed after the layout was fixed is an error, not a silent no-op
    pb = [torch.nn.Parameter(t.clone().to(d)) for t in init]
    ob = harness.FlatSGD(pb, lr=0.1)
    for p, gr in zip(pb, grads[0]):
        p.grad = gr.clone().to(d)
    ob.step()
    pb[0].data = pb[0].data.clone()
    with pytest.raises(RuntimeError):
        ob.step()


def test_dropout_seed_reproducible_and_fresh_per_forward():
    """the dropout key comes from torch's CPU generator (reproducible under manual_seed, new on every
    forward, also on throw-away nn.DataParallel-style replicas), not from id(self) / a per-object call counter."""
    from torch.nn.parallel import replicate
    m, nc = _model('st_gcnold', torch.float32, dropout=0.5)
    m.train()
    x = det_tensor('seed.x', (2, 3, 32, 25, 2)).to(dev())

    def run(mod):
        with torch.no_grad():
            return mod(x).clone()
    torch.manual_seed(7)
    a1, a2 = run(m), run(m)
    torch.manual_seed(7)
    b1 = run(m)
    # (the BatchNorm batch sums are fp64 atomics: two runs with the SAME masks agree to round-off, not bit for bit)
    assert rel_err(a1, a2) > 1e-2             # consecutive forwards: different masks
    assert rel_err(a1, b1) < 1e-5             # same seed: same masks
    # replicas are rebuilt for every forward (DataParallel): their masks must still change from call to call
    torch.manual_seed(9)
    r1 = run(replicate(m, [0])[0])
    r2 = run(replicate(m, [0])[0])
    assert rel_err(r1, r2) > 1e-2


def test_nnz_cap_overflow_is_detected(ops):
    """a reduced nnz_cap smaller than nnz(A) truncates the in-LDS lists: with ops.CHECK_NNZ the kernel's status flag
    turns that into an error instead of a silently wrong y."""
    d = dev()
    V, K, C = 25, 3, 16
    A = torch.rand(K, V, V, device=d) + 0.1                    # dense: nnz = 1875
    x = torch.randn(2, 8, V, C, device=d)
    wp = ops.pack_gcn_weight(torch.randn(C, K, C, device=d), torch.float32)
    ops.CHECK_NNZ = True
    try:
        ops.gcn_forward(x, A, wp, C, nnz_cap=K * V * V)        # full capacity: fine
        with pytest.raises(RuntimeError):
            ops.gcn_forward(x, A, wp, C, nnz_cap=100)
    finally:
        ops.CHECK_NNZ = False


def test_adjacency_gradient_pattern_keeps_zero_importance_entries(ops):
    """dA is taken on the constant adjacency pattern: an importance value that is exactly 0 still receives the gradient
    autograd of tgcn.py:86 gives it (it used to be dropped because A_eff there is 0)."""
    d = dev()
    from istgcn_amd.net.utils.graph import Graph
    B = torch.tensor(Graph('ntu-rgb+d', 'spatial').A, dtype=torch.float32)
    K, V, _ = B.shape
    g = torch.Generator().manual_seed(4)
    imp = torch.rand(K, V, V, generator=g) + 0.5
    nzidx = (B != 0).nonzero()
    for i in nzidx[::7]:
        imp[tuple(i)] = 0.0                                    # exact zeros ON the pattern
    NM, T, C = 2, 6, 16
    x = torch.randn(NM, T, V, C, generator=g)
    dy = torch.randn(NM, T, V, C, generator=g)
    W3 = torch.randn(K, C, C, generator=g) * C ** -0.5
    A_eff = (B * imp).requires_grad_(True)
    h = torch.einsum('kci,ntvi->nktvc', W3, x)
    y = torch.einsum('nktvc,kvw->ntwc', h, A_eff)
    y.backward(dy)
    want = A_eff.grad * (B != 0)
    pat = (B != 0).float().to(d)
    cap = int(pat.sum())
    _, dA = ops.gcn_bwd_data(dy.to(d), (B * imp).to(d), W3.to(d), x=x.to(d), want_dA=True, nnz_cap=cap, pattern=pat)
    assert diag('dA_pattern', dA, want, 3e-5) < 3e-5
    zero_on_pattern = ((B != 0) & (imp == 0))
    assert zero_on_pattern.any() and float(want[zero_on_pattern].abs().max()) > 0
    # dense pattern (learnable dense A through the unit drop-in): every entry, as autograd
    _, dA2 = ops.gcn_bwd_data(dy.to(d), (B * imp).to(d), W3.to(d), x=x.to(d), want_dA=True, nnz_cap=K * V * V,
                              pattern=torch.ones(K, V, V, device=d))
    assert diag('dA_dense', dA2, A_eff.grad, 3e-5) < 3e-5


def test_model_eval_logits_absolute_error(golden):
    """north_star states the bar as an absolute 1e-3 on the logits; the G4 fixtures reach |logit| ~ 100-150, so the
    relative gate above admits ~0.1 absolute.  Assert the absolute bound as well and record the measured figures."""
    import os
    from gpu_util import OUT
    lines = []
    for tag in ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn', 'st_gcn_mstcn_1x1_deep']:
        g = golden('model_g4_%s.npz' % tag)
        m, nc = _model(tag, torch.float32)
        m.eval()
        x = det_tensor('g4.x.' + tag, tuple(int(s) for s in g['eval_shape'])).to(dev())
        with torch.no_grad():
            y = m(x).double().cpu()
        ref = torch.from_numpy(g['eval_logits']).double()
        err = float((y - ref).abs().max())
        lines.append('%-28s max|logit| %8.3f  max abs err %.3e  rel %.3e' % (tag, float(ref.abs().max()), err,
                                                                          err / float(ref.abs().max())))
        assert err < 1e-3, lines[-1]
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'parity_abs_logits.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


@pytest.mark.parametrize('tag', ['st_gcn_mstcn_1x1_deep', 'st_gcn_msgcn'])
def test_model_fp16_storage_close_to_fp32(golden, tag):
    """BASELINE config 5 = net/st_gcn_mstcn_1x1_deep.py:253-269 in float16 at T=600: float16 activations (fp32
    accumulation, fp64 statistics, static loss scale for the backward pass) against the reference's fp32 logits and the
    fp32 HIP gradients.  float16 carries 3 more mantissa bits than bfloat16: logits within 5e-3 relative L2."""
    g = golden('model_g4_%s.npz' % tag)
    shp = tuple(int(s) for s in g['train_shape'])
    LS = 4096.0
    grads = {}
    for dt in (torch.float32, torch.float16):
        m, nc = _model(tag, dt)
        x = det_tensor('g4.xt.' + tag, shp).to(dev())
        lab = det_labels('g4.lab.' + tag, shp[0], nc).to(dev())
        m.train()
        logits = m(x)
        loss = F.cross_entropy(logits, lab)
        (loss * (LS if dt == torch.float16 else 1.0)).backward()
        sc = 1.0 / LS if dt == torch.float16 else 1.0
        grads[dt] = ([None if p.grad is None else p.grad.double().flatten() * sc for p in m.parameters()],
                     logits.detach(), float(loss))
    from gpu_util import gate16
    g32, g16 = grads[torch.float32], grads[torch.float16]
    gl, gg = FP16_MODEL_GATE.get(tag, (5e-3, 0.08))
    assert gate16('model_fp16 %s train logits rel-L2 vs reference' % tag, l2rel(g16[1], g['train_logits']), gl)
    assert abs(g16[2] - float(g['train_loss'])) < 1e-2
    a = torch.cat([t for t in g32[0] if t is not None])
    b = torch.cat([t for t in g16[0] if t is not None])
    assert torch.isfinite(b).all()
    assert gate16('model_fp16 %s whole-gradient rel-L2 vs fp32 HIP' % tag, float((a - b).norm() / a.norm()), gg)
    # eval logits at the full clip shape (T=600 for the deep model)
    m, nc = _model(tag, torch.float16)
    m.eval()
    x = det_tensor('g4.x.' + tag, tuple(int(s) for s in g['eval_shape'])).to(dev())
    with torch.no_grad():
        y = m(x)
    assert l2rel(y, g['eval_logits']) < 5e-3


# 16-bit whole-model gates per model: (logits rel-L2, whole-gradient rel-L2[, worst 1 - cosine]); measured values are
# appended to gpurun_out/err16_measured.txt by every run, the gates sit at <= 2x those
# measured round 3 (bf16: logits .0025/.0023/.0024, gradient .098/.172/.121, 1-cos .060/.088/.098; fp16: logits 4.0e-4/2.8e-4,
# gradient .055/.036)
BF16_MODEL_GATE = {'st_gcn_msgcn': (5e-3, 0.2, 0.125), 'st_gcn_multi3_fix_3A_mstcn': (5e-3, 0.25, 0.18),
                   'st_gcn_mstcn_1x1': (5e-3, 0.25, 0.4)}
# (norm-weighted mean, 12 runs each: 0.0180-0.0183 / 0.0349-0.0357 / 0.0343-0.0370 -> gates at 2x)
BF16_MODEL_COS_MEAN_GATE = {'st_gcn_msgcn': 0.036, 'st_gcn_multi3_fix_3A_mstcn': 0.07, 'st_gcn_mstcn_1x1': 0.072}
FP16_MODEL_GATE = {'st_gcn_mstcn_1x1_deep': (8e-4, 0.08), 'st_gcn_msgcn': (6e-4, 0.072)}
FULL_BATCH = [('st_gcn_msgcn', 64, 300, torch.bfloat16), ('st_gcn_msgcn', 64, 300, torch.float16),
              ('st_gcn_multi3_fix_3A_mstcn', 64, 300, torch.bfloat16), ('st_gcn_mstcn_1x1', 256, 300, torch.bfloat16)]
# measured (gpurun_out/err16_measured.txt, round 3): logits rel-L2 1.2e-3 / 1.5e-4 / 1.1e-3 / 1.2e-3, whole-gradient
# rel-L2 .089 / .032 / .098 / .158 -> gates at <= 2x
FULL_BATCH_GATE = {('st_gcn_msgcn', torch.bfloat16): (2.4e-3, 0.18), ('st_gcn_msgcn', torch.float16): (3.1e-4, 0.064),
                   ('st_gcn_multi3_fix_3A_mstcn', torch.bfloat16): (2.2e-3, 0.2), ('st_gcn_mstcn_1x1', torch.bfloat16): (2.4e-3, 0.3)}


@pytest.mark.parametrize('tag,batch,T,dt', FULL_BATCH, ids=['%s_b%d_%s' % (t, b, str(d_)[6:]) for t, b, _, d_ in FULL_BATCH])
def test_full_batch_step_16bit_vs_fp32_hip(tag, batch, T, dt):
    """BASELINE configs 2, 4 and 3 at their real sizes -- st_gcn_msgcn and st_gcn_multi3_fix_3A_mstcn with 64 clips (NM =
    128), st_gcn_mstcn_1x1 on 18-joint skeletons with 256 clips (NM = 512), T = 300 -- one training step in 16-bit
    storage against the SAME step of the fp32 HIP path (itself pinned to the reference by the G4 tests): the tests that
    run whole models with every workgroup of every kernel resident, apart from bench.py."""
    from istgcn_amd import harness
    from gpu_util import gate16
    gargs, nc = MODEL_CFG[tag]
    Vj = 18 if gargs['layout'] == 'openpose' else 25
    mod = importlib.import_module('istgcn_amd.net.' + tag)
    res = {}
    for d_ in (torch.float32, dt):
        torch.manual_seed(0)
        m = mod.Model(3, nc, gargs, True, dropout=0, compute_dtype=d_)
        m.apply(harness.weights_init)
        m.to(dev()).train()
        gen = torch.Generator().manual_seed(11)
        x = torch.randn(batch, 3, T, Vj, 2, generator=gen).to(dev())
        y = torch.randint(0, nc, (batch,), generator=gen).to(dev())
        ls = 65536.0 if d_ == torch.float16 else 1.0
        logits = m(x)
        loss = F.cross_entropy(logits, y)
        (loss * ls).backward()
        res[d_] = (logits.detach().double().cpu(), float(loss),
                   [None if p.grad is None else (p.grad.double() / ls).flatten().cpu() for p in m.parameters()])
        del m, x, logits, loss
        torch.cuda.empty_cache()
    (l32, loss32, g32), (l16, loss16, g16) = res[torch.float32], res[dt]
    tol_l, tol_g = FULL_BATCH_GATE[(tag, dt)]
    name = 'full_batch_step %s b%d %s' % (tag, batch, str(dt)[6:])
    assert gate16(name + ' logits rel-L2 vs fp32 HIP', float((l16 - l32).norm() / l32.norm()), tol_l)
    assert abs(loss16 - loss32) < 2e-2
    a = torch.cat([t for t in g32 if t is not None])
    b = torch.cat([t for t in g16 if t is not None])
    assert torch.isfinite(b).all()
    assert gate16(name + ' whole-gradient rel-L2 vs fp32 HIP', float((a - b).norm() / a.norm()), tol_g)


@pytest.mark.gpu
def test_graphed_step_matches_eager_and_redraws_dropout(ops):
    """harness.GraphedStep (forward + backward replayed from one hipGraph, SGD eager) == eager train_step when dropout is
    off; with dropout the device-side seed epoch gives every replay its own mask (a frozen host seed would repeat it)."""
    import copy
    from istgcn_amd import harness
    from istgcn_amd.net import st_gcn_msgcn as prod
    d = torch.device('cuda:0')
    gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 32, 25, 2, generator=g).to(d)
    y = torch.randint(0, 60, (4,), generator=g).to(d)
    torch.manual_seed(0)
    m0 = prod.Model(3, 60, gargs, True, dropout=0).to(d).train()
    m1 = copy.deepcopy(m0)
    o0, o1 = harness.make_optimizer(m0), harness.make_optimizer(m1)
    for o in (o0, o1):
        o.param_groups[0]['lr'] = 0.01
    # GraphedStep runs two eager steps itself before capturing: give the eager twin the same head start
    for _ in range(2):
        harness.train_step(m0, o0, x, y)
    gs = harness.GraphedStep(m1, o1, x, y, warmup=2)
    for _ in range(3):
        l0 = harness.train_step(m0, o0, x, y)
        l1 = gs(x, y)
    torch.cuda.synchronize()
    assert abs(float(l0) - float(l1)) < 2e-3 * max(1.0, abs(float(l0)))
    w0 = torch.cat([p.detach().flatten() for p in m0.parameters()])
    w1 = torch.cat([p.detach().flatten() for p in m1.parameters()])
    assert rel_err(w1, w0) < 1e-3
    # dropout on, learning rate 0: identical weights and inputs every step, so the loss moves only with the mask
    m2 = prod.Model(3, 60, gargs, True, dropout=0.5).to(d).train()
    o2 = harness.make_optimizer(m2)
    o2.param_groups[0]['lr'] = 0.0
    o2.param_groups[0]['weight_decay'] = 0.0
    gs2 = harness.GraphedStep(m2, o2, x, y, warmup=1)
    losses = [float(gs2(x, y)) for _ in range(4)]
    assert len({round(v, 5) for v in losses}) > 1, losses
    assert int(m2.device_seed_epoch().item()) >= 4


@pytest.mark.gpu
def test_dataparallel_wrapper_runs_the_hip_path(ops):
    """the reference wraps the model in nn.DataParallel (processor/my_io.py:86-87).  With the one visible
    GPU as its only replica the wrapper must give the bare model's logits and gradients (kernels take the current stream
    of the tensor's device and keep no global state that a replica could trip over)."""
    from istgcn_amd.net import st_gcn_msgcn as prod
    d = torch.device('cuda:0')
    gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
    torch.manual_seed(1)
    m = prod.Model(3, 60, gargs, True, dropout=0).to(d).train()
    x = torch.randn(4, 3, 24, 25, 2, generator=torch.Generator().manual_seed(2)).to(d)
    y = torch.randint(0, 60, (4,), generator=torch.Generator().manual_seed(3)).to(d)
    out0 = m(x)
    torch.nn.functional.cross_entropy(out0, y).backward()
    g0 = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone()
    m.zero_grad()
    dpm = torch.nn.DataParallel(m, device_ids=[0])
    out1 = dpm(x)
    torch.nn.functional.cross_entropy(out1, y).backward()
    g1 = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert rel_err(out1, out0) < 1e-5
    assert rel_err(g1, g0) < 2e-3          # (fp32 atomics + isolated ReLU flips between two runs of the same path)


@pytest.mark.gpu
def test_two_replicas_from_two_threads_on_two_streams(ops):
    """what nn.DataParallel does per device (processor/my_io.py:86-87: a device LIST) on the one GPU a test
    box has -- two replicas of one model driven from two host threads on two streams: per-thread scratch and BatchNorm-tail
    arming, a StepArena per replica, the weight-pack plan shared through the source module.  Each replica must give what
    the bare model gives on its half of the batch (train mode: BatchNorm statistics per replica, as under DataParallel), and
    ONE backward over both losses must give the sum of the two halves' gradients.  Also: a replica whose parameters are
    fresh tensors (another device's copy) re-points the shared plan instead of building one."""
    import threading
    from torch.nn.parallel import replicate
    from istgcn_amd.net import st_gcn_msgcn as prod
    d = torch.device('cuda:0')
    gargs = dict(layout='ntu-rgb+d', strategy='spatial_3')
    torch.manual_seed(4)
    m = prod.Model(3, 60, gargs, True, dropout=0).to(d).train()
    m.load_state_dict(det_fill_(m.state_dict()))
    x = torch.randn(4, 3, 24, 25, 2, generator=torch.Generator().manual_seed(5)).to(d)
    y = torch.randint(0, 60, (4,), generator=torch.Generator().manual_seed(6)).to(d)
    halves = [(x[:2], y[:2]), (x[2:], y[2:])]
    ref_out, ref_grad = [], {}
    for xh, yh in halves:                                   # the bare model on each half
        m.zero_grad()
        o = m(xh)
        F.cross_entropy(o, yh).backward()
        for n_, p in m.named_parameters():
            if p.grad is not None:
                ref_grad[n_] = ref_grad[n_] + p.grad if n_ in ref_grad else p.grad.clone()
        ref_out.append(o.detach().clone())
    m.zero_grad()
    assert len(m._pack_plans) == 1
    plan_before = next(iter(m._pack_plans.values()))[1]
    reps = [replicate(m, [0])[0] for _ in halves]
    streams = [torch.cuda.Stream(device=d) for _ in halves]
    outs, errs = [None, None], []

    def work(i):
        try:
            with torch.cuda.device(d), torch.cuda.stream(streams[i]):
                streams[i].wait_stream(torch.cuda.default_stream(d))
                outs[i] = reps[i](halves[i][0])
        except Exception:           # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, '\n'.join(errs)
    for s_ in streams:
        torch.cuda.current_stream(d).wait_stream(s_)
    loss = sum(F.cross_entropy(outs[i], halves[i][1]) for i in range(2))
    loss.backward()
    torch.cuda.synchronize()
    for i in range(2):
        assert rel_err(outs[i], ref_out[i]) < 1e-5
    got = {n_: p.grad for n_, p in m.named_parameters() if p.grad is not None}
    assert set(ref_grad) <= set(got)
    scale = max(float(v.abs().max()) for v in ref_grad.values())
    for n_, v in got.items():
        if n_ in ref_grad:
            assert float((v - ref_grad[n_]).abs().max()) < 2e-3 * max(1.0, scale), n_
        else:       # the reference's dead parameters (gcn.branch.bn.*): Broadcast's backward hands them zeros instead of None
            assert float(v.abs().max()) == 0.0, n_
    assert len(m._pack_plans) == 1 and next(iter(m._pack_plans.values()))[1] is plan_before     # the replicas used the source's plan
    # (a replica's parameters are fresh tensors on every forward -- Broadcast copies them even onto the source's own device --
    #  so each of the forwards above re-pointed the shared plan; once more, alone)
    o = replicate(m, [0])[0](halves[0][0])
    assert rel_err(o, ref_out[0]) < 1e-5
    assert len(m._pack_plans) == 1 and next(iter(m._pack_plans.values()))[1] is plan_before


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(6, 2, 75, 25, 256), (4, 1, 13, 18, 64), (2, 2, 5, 25, 20), (3, 1, 2, 3, 7)])
def test_pooling_matches_avg_pool_and_person_mean(ops, shape, dt):
    """PoolFn = F.avg_pool2d over (T, V) + mean over the M persons (net/st_gcnold.py:89-91), forward and backward."""
    from istgcn_amd import functional as Fn
    N, M, T, V, C = shape
    g = torch.Generator().manual_seed(5)
    y = torch.randn(N * M, T, V, C, generator=g).to(dt)
    w = torch.randn(N, C, generator=g)
    yr = y.float().clone().requires_grad_(True)
    ref = yr.mean(dim=(1, 2)).view(N, M, C).mean(dim=1)
    (ref * w).sum().backward()
    yd = y.to(dev()).detach().requires_grad_(True)
    feat = Fn.PoolFn.apply(yd, M)
    (feat * w.to(dev())).sum().backward()
    torch.cuda.synchronize()
    assert feat.dtype == torch.float32 and feat.shape == (N, C)
    assert (feat.cpu() - ref.detach()).abs().max() < 2e-6 * max(1.0, float(T * V) ** 0.5)
    want = yr.grad.to(dt).float()
    assert (yd.grad.float().cpu() - want).abs().max() <= 1e-6 + 2.0 ** -7 * want.abs().max() * (0 if dt == torch.float32 else 1)


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('C,with_res', [(64, False), (256, True), (24, False)])
def test_block_out_relu_mask_equals_reading_out(ops, C, with_res, dt):
    """The forward's one-byte-per-vector ReLU mask makes the backward bit-identical to the one that reads `out`
    (st_gcnold.py:201-203: relu(tcn(x) + res)); channel counts without a vector map get no mask."""
    rows = 3 * 7 * 25
    g = torch.Generator().manual_seed(C)
    d = dev()
    z = torch.randn(3, 7, 25, C, generator=g).to(d, dt)
    res = torch.randn(3, 7, 25, C, generator=g).to(d, dt) if with_res else None
    coef = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2,
                        torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).to(d)
    coefr = coef.flip(1).contiguous() if with_res else None
    out, rmask = ops.block_out_fwd(z, coef[:2].contiguous(), res, None if coefr is None else coefr[:2].contiguous(), 0.3, 11,
                                   want_mask=True)
    assert (rmask is not None) == ops.relu_mask_ok(C, dt)
    if rmask is None:
        return
    epl = 4 if dt == torch.float32 else 8
    want = ((out.view(-1, epl) > 0).to(torch.int32) << torch.arange(epl, device=d, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(rmask, want)
    dout = torch.randn(3, 7, 25, C, generator=g).to(d, dt)
    a = ops.block_out_bwd(dout, out, z, coef, res, coefr, 0.3, 11)
    b = ops.block_out_bwd(dout, None, z, coef, res, coefr, 0.3, 11, relu_mask=rmask)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and rows * C == out.numel()
    assert (a[1] - b[1]).abs().max() <= 1e-9 * a[1].abs().max() + 1e-12
    if with_res:
        assert (a[2] - b[2]).abs().max() <= 1e-9 * a[2].abs().max() + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_dres_is_never_needed_as_a_tensor(ops, dt):
    """Round 5: the block's backward does not write dres = dout * [out > 0] (net/st_gcnold.py:201-203) any more -- every reader
    takes dout and the forward's byte mask.  Each reader's masked form is bit-identical to its form on the dres tensor:
    block_out_bwd's sums without the store, affine2 (tcn.3's / the residual BatchNorm's backward, with the dropout mask), and
    the identity-residual addend of the graph conv's data gradient (register-chained kernel: 16-bit storage)."""
    from istgcn_amd.net.utils.graph import Graph
    d = dev()
    g = torch.Generator().manual_seed(5)
    NM, T, V = 3, 9, 25
    for C in (64, 128, 256):
        z = torch.randn(NM, T, V, C, generator=g).to(d, dt)
        res = torch.randn(NM, T, V, C, generator=g).to(d, dt)
        coef = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2,
                            torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).to(d)
        out, rmask = ops.block_out_fwd(z, coef[:2].contiguous(), res, None, 0.3, 11, want_mask=True)
        dout = torch.randn(NM, T, V, C, generator=g).to(d, dt)
        dres, st_a, _ = ops.block_out_bwd(dout, None, z, coef, None, None, 0.3, 11, relu_mask=rmask)
        none, st_b, _ = ops.block_out_bwd(dout, None, z, coef, None, None, 0.3, 11, relu_mask=rmask, want_dres=False)
        assert none is None and (st_a - st_b).abs().max() <= 1e-9 * st_a.abs().max()
        assert torch.equal(dres, torch.where(out > 0, dout, torch.zeros_like(dout)))
        abc = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1]).to(d)
        for p_drop in (0.0, 0.3):
            a = ops.affine2(dres, z, abc, p_drop, 11)
            b = ops.affine2(dout, z, abc, p_drop, 11, relu_mask=rmask)
            assert torch.equal(a, b), (C, p_drop)
        if dt == torch.float32:
            assert not ops.gcn_bwd_addend_mask_ok(V, C, C, 3, dt)
            continue
        assert ops.gcn_bwd_addend_mask_ok(V, C, C, 3, dt)
        gr = Graph('ntu-rgb+d', 'spatial_3')
        A = torch.tensor(gr.A + gr.A2 + gr.A3, dtype=torch.float32, device=d)
        W3 = (torch.randn(3, C, C, generator=g) * C ** -0.5).to(d)
        dy = torch.randn(NM, T, V, C, generator=g).to(d, dt)
        x = torch.randn(NM, T, V, C, generator=g).to(d, dt)
        for want_dA in (True, False):
            with ops.trace() as tr:
                dx_a, dA_a = ops.gcn_bwd_data(dy, A, W3, x=x, addend=dres, want_dA=want_dA)
                dx_b, dA_b = ops.gcn_bwd_data(dy, A, W3, x=x, addend=dout, addend_mask=rmask, want_dA=want_dA)
            assert tr.ran('gcn_rc_bwd_kernel') and not tr.ran('_114gcn_bwd_kernel')      # (mangled: not the packer's name)
            assert torch.equal(dx_a, dx_b), (C, want_dA)
            if want_dA:
                assert (dA_a - dA_b).abs().max() <= 1e-5 * dA_a.abs().max()


@pytest.mark.gpu
def test_flat_sgd_tap_major_layout_and_checkpoint_round_trip(tmp_path):
    """FlatSGD stores the temporal-conv weights [k][Cout][Cin] inside its flat buffers (the layout their gradient is computed
    in; the parameter keeps the Conv2d shape).  Updates must equal torch.optim.SGD's on a contiguous twin, gradients must
    arrive without a layout copy, and a checkpoint (torchlight io.py:101-107 / 57-90) must round-trip through plain tensors."""
    from istgcn_amd import harness
    d = dev()
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(16, 16, 9, 1, generator=g)
    b0 = torch.randn(16, generator=g)
    mk = lambda: (torch.nn.Parameter(w0.clone().to(d)), torch.nn.Parameter(b0.clone().to(d)))  # noqa: E731
    (wa, ba), (wb, bb) = mk(), mk()
    wa._istgcn_flat_layout = 'tap_major'
    oa = harness.FlatSGD([wa, ba], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    ob = torch.optim.SGD([wb, bb], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    for step in range(3):
        oa.zero_grad()
        ob.zero_grad()
        gw = torch.randn(9, 16, 16, generator=g).to(d)                 # what tconv_wgrad produces: [k][Cout][Cin]
        gb = torch.randn(16, generator=g).to(d)
        wa.grad, ba.grad = gw.permute(1, 2, 0).unsqueeze(-1), gb.clone()
        wb.grad, bb.grad = gw.permute(1, 2, 0).unsqueeze(-1).contiguous(), gb.clone()
        oa.step()
        ob.step()
        assert rel_err(wa, wb) < 2e-6 and rel_err(ba, bb) < 2e-6
    assert wa.shape == (16, 16, 9, 1) and not wa.is_contiguous()
    assert wa.data.view(16, 16, 9).permute(2, 0, 1).is_contiguous()    # what the packers and the gradient see
    assert wa.grad.stride()[:3] == wa.stride()[:3]
    mod = torch.nn.Module()
    mod.w, mod.b = wa, ba
    path = harness.save_model(mod, str(tmp_path / 'm.pt'))
    sd = torch.load(path)
    assert all(v.is_contiguous() for v in sd.values())
    fresh = torch.nn.Module()
    fresh.w, fresh.b = torch.nn.Parameter(torch.zeros(16, 16, 9, 1)), torch.nn.Parameter(torch.zeros(16))
    harness.load_weights(fresh, path)
    assert torch.equal(fresh.w.detach(), wa.detach().cpu()) and torch.equal(fresh.b.detach(), ba.detach().cpu())


@pytest.mark.gpu
@pytest.mark.parametrize('tag,dt', [('st_gcn_msgcn', torch.bfloat16), ('st_gcnold', torch.float32), ('st_gcn_mstcn_1x1', torch.float16),
                                    ('st_gcn_multi3_fix_3A_mstcn', torch.bfloat16)])
def test_bn_tails_equal_standalone_launches(tag, dt):
    """"Last workgroup finalises" (csrc/bn_tail.hpp): two training steps with the BatchNorm arithmetic carried as the tail of the
    kernels that produce the batch sums against the same steps with the stand-alone bn_finalize / bn_bwd_coef launches
    (ops.BN_TAILS = False): logits, running statistics and gradients agree to the noise of the fp64 atomics' summation
    order; the tails really are taken where the kernel variants have them, and the sums / tickets are left zeroed (the
    second step would otherwise double-count)."""
    from istgcn_amd import harness, ops
    gargs, nc = MODEL_CFG[tag]
    Vj = 18 if gargs['layout'] == 'openpose' else 25
    mod = importlib.import_module('istgcn_amd.net.' + tag)
    res = {}
    keep = ops.BN_TAILS
    try:
        for key, flag in (('tail', True), ('alone', False), ('alone2', False)):
            ops.BN_TAILS = flag
            ops.TAIL_STATS['taken'] = ops.TAIL_STATS['standalone'] = 0
            torch.manual_seed(0)
            m = mod.Model(3, nc, gargs, True, dropout=0, compute_dtype=dt)
            m.apply(harness.weights_init)
            m.to(dev()).train()
            gen = torch.Generator().manual_seed(5)
            x = torch.randn(4, 3, 48, Vj, 2, generator=gen).to(dev())
            y = torch.randint(0, nc, (4,), generator=gen).to(dev())
            ls = 1024.0 if dt == torch.float16 else 1.0
            for _ in range(2):
                m.zero_grad()
                logits = m(x)
                (F.cross_entropy(logits, y) * ls).backward()
            rs = torch.cat([b.detach().double().flatten() for n_, b in m.named_buffers() if 'running' in n_]).cpu()
            gr = torch.cat([(p.grad.double() / ls).flatten() for p in m.parameters() if p.grad is not None]).cpu()
            res[key] = (logits.detach().double().cpu(), rs, gr, dict(ops.TAIL_STATS))
    finally:
        ops.BN_TAILS = keep
    (l1, r1, g1, st1), (l0, r0, g0, st0), (l2, r2, g2, _) = res['tail'], res['alone'], res['alone2']
    assert st0['taken'] == 0 and st1['taken'] > 0, (st0, st1)
    if dt == torch.bfloat16:
        assert st1['taken'] >= 2 * 30, st1                       # 16-bit trunk: every block's four BatchNorm sites
    rel = lambda a, b: float((a - b).norm() / max(1e-30, float(b.norm())))
    # the yardstick is the run-to-run difference of the stand-alone path itself (the order of the fp64 / fp32 atomics varies;
    # a pre-activation within round-off of zero then takes the other ReLU branch, and in 16-bit storage a 1e-5 change of a
    # BatchNorm coefficient re-draws the rounding of a fraction of the stored activations)
    from gpu_util import OUT
    import os
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'bn_tails_measured.txt'), 'a') as f:
        f.write('%s %s: tails taken %d stand-alone %d | tail vs alone: running %.3g logits %.3g grads %.3g | alone vs alone: running %.3g '
                'logits %.3g grads %.3g\n' % (tag, str(dt)[6:], st1['taken'], st1['standalone'], rel(r1, r0), rel(l1, l0), rel(g1, g0),
                                              rel(r2, r0), rel(l2, l0), rel(g2, g0)))
    # (16-bit storage: two stand-alone runs differ by 1e-6 .. 2e-5 in the running statistics themselves -- gpurun_out/
    #  bn_tails_measured.txt: 1.4e-5 for st_gcn_msgcn, 1.3e-6 for the same test of st_gcn_multi3 on another day -- so the
    #  yardstick alone is not a bound; the floor is the larger of those)
    assert rel(r1, r0) < 3 * rel(r2, r0) + (1e-5 if dt == torch.float32 else 5e-5), (rel(r1, r0), rel(r2, r0))
    assert rel(l1, l0) < 3 * rel(l2, l0) + (1e-5 if dt == torch.float32 else 5e-3), (rel(l1, l0), rel(l2, l0))
    # (float32: logits agree to 3e-7, but one ReLU mask flip at a pre-activation within round-off of zero moves the whole
    #  gradient by 1e-3 .. 4e-3 -- discrete events, so two runs of the SAME path differ by 0.0006 .. 0.002 as well; cf. smoke())
    assert torch.isfinite(g1).all() and rel(g1, g0) < 3 * rel(g2, g0) + (1e-2 if dt == torch.float32 else 0.1), (rel(g1, g0), rel(g2, g0))
