"""CPU: the oracle (oracle/) against the golden fixtures produced by the reference itself.
This is what pins the oracle; the HIP product is then checked against
both the oracle and the same fixtures in the -m gpu tests."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from detinit import (det_fill_, det_tensor, det_labels, subsample, WIDE_UNITS, WIDE_BLOCKS, wide_unit_inputs, wide_unit_names,
                     wide_block_inputs, wide_block_has)
from oracle.graph_ref import GraphRef
from oracle import stgcn_ref as R

MAN = json.load(open(os.path.join(GOLDEN, 'graph_g1.json')))
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


@pytest.mark.parametrize('tag', sorted(MAN))
def test_graph_bit_exact(tag, golden):
    g1 = golden('graph_g1.npz')
    lay, st = tag.split('|')
    rec = MAN[tag]
    if 'error' in rec:
        with pytest.raises(Exception) as ei:
            GraphRef(lay, st)
        assert type(ei.value).__name__ == rec['error']
        return
    g = GraphRef(lay, st)
    assert g.num_node == rec['num_node'] and g.center == rec['center']
    assert np.array_equal(np.asarray(g.edge), g1[tag + '|edge'])
    assert np.array_equal(g.A, g1[tag + '|A'])          # float64, bit exact
    if rec['has_A23']:
        assert np.array_equal(g.A2, g1[tag + '|A2']) and np.array_equal(g.A3, g1[tag + '|A3'])


def test_graph_max_hop_arg(golden):
    assert np.array_equal(GraphRef('ntu-rgb+d', 'spatial', max_hop=2).A, golden('graph_g1.npz')['ntu-rgb+d|spatial|max_hop2|A'])


def _unit_case(g, ci):
    b = 'c%d.' % ci
    t = lambda k: torch.from_numpy(g[b + k])  # noqa: E731
    return b, t


@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('unit', ['tgcn', '3a', 'inc', 'incnew', 'multi3', 'multi3fix', 'only3'])
def test_gcn_units(unit, ci, golden):
    g = golden('units_g2.npz')
    b, t = _unit_case(g, ci)
    x, r, W, bias = t('x'), t('r'), t('W'), t('b')
    A, A2, A3 = t('A'), t('A2'), t('A3')
    K, cout, cin = A.shape[0], W.shape[0] // A.shape[0], W.shape[1]
    imps = [t('imp%d' % j).clone().requires_grad_(True) for j in (1, 2, 3)]
    kind = {'tgcn': 'plain', '3a': '3a', 'inc': 'incep', 'incnew': 'incep'}.get(unit, 'plain')
    u = R.RefGCN(kind, cin, cout, K)
    conv = u.branch.conv if kind == 'incep' else u.conv
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(bias)
    xx = x.clone().requires_grad_(True)
    if unit == 'tgcn':
        adj = (A * imps[0],)
    elif unit == '3a':
        adj = (A, imps[0], imps[1], imps[2])
    elif unit in ('inc', 'incnew'):
        adj = (A * imps[0], A2 * imps[1], A3 * imps[2])
    else:  # folded-adjacency variants of SURVEY 2.1 #11: one kernel, different A_eff
        Ai = A * imps[0]
        adj = ({'multi3': Ai + Ai ** 2 + Ai ** 3, 'multi3fix': (Ai + Ai ** 2 + Ai ** 3) / 3, 'only3': Ai ** 3}[unit],)
    y = u(xx, adj)
    (y * r).sum().backward()
    assert rel_err(y, g[b + unit + '.y']) < 1e-5
    assert rel_err(xx.grad, g[b + unit + '.dx']) < 1e-5
    if unit + '.dW' in ''.join(k for k in g.files if k.startswith(b)):
        assert rel_err(conv.weight.grad, g[b + unit + '.dW']) < 1e-5
        assert rel_err(conv.bias.grad, g[b + unit + '.db']) < 1e-5
    for j in (1, 2, 3):
        key = b + unit + '.dimp%d' % j
        if key in g.files:
            assert rel_err(imps[j - 1].grad, g[key]) < 1e-5


def _block_adj(kind, A, A2, A3, imps):
    gk = R.KINDS[kind][0]
    if gk == 'plain':
        return (A * imps[0],)
    if gk == 'incep':
        return (A * imps[0], A2 * imps[1], A3 * imps[2])
    return (A, imps[0], imps[1], imps[2])


@pytest.mark.parametrize('kind', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn'])
def test_blocks(kind, golden):
    g = golden('block_g3_%s.npz' % kind)
    bases = sorted({'.'.join(k.split('.')[:2]) + '.' for k in g.files})
    assert len(bases) == 4
    for b in bases:
        t = lambda k: torch.from_numpy(g[b + k])  # noqa: E731
        sd = {k[len(b) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(b + 'sd.')}
        A, A2, A3, x, r = t('A'), t('A2'), t('A3'), t('x'), t('r')
        cout, cin = sd['gcn.conv.weight' if 'gcn.conv.weight' in sd else 'gcn.branch.conv.weight'].shape[:2]
        cout //= A.shape[0]
        stride = x.shape[2] // r.shape[2]
        residual = b.split('.')[1] != 's0'
        blk = R.RefBlock(kind, cin, cout, A.shape[0], stride, dropout=0, residual=residual)
        blk.load_state_dict(sd, strict=True)
        imps = [t('imp%d' % j).clone().requires_grad_(True) for j in (1, 2, 3)]
        mst = t('mst').clone().requires_grad_(True)
        blk.eval()
        with torch.no_grad():
            assert rel_err(blk(x, _block_adj(kind, A, A2, A3, imps), mst), g[b + 'y_eval']) < 1e-5
        blk.train()
        xx = x.clone().requires_grad_(True)
        y = blk(xx, _block_adj(kind, A, A2, A3, imps), mst)
        (y * r).sum().backward()
        assert rel_err(y, g[b + 'y_train']) < 1e-5
        assert rel_err(xx.grad, g[b + 'dx']) < 1e-4
        n_grad = 0
        for k, p in blk.named_parameters():
            if b + 'grad.' + k in g.files:
                assert rel_err(p.grad, g[b + 'grad.' + k]) < 1e-4, k
                n_grad += 1
            else:
                assert p.grad is None, k      # dead params stay grad-less, as upstream
        assert n_grad > 0
        for j in (1, 2, 3):
            if b + 'dimp%d' % j in g.files:
                assert rel_err(imps[j - 1].grad, g[b + 'dimp%d' % j]) < 1e-4
        if b + 'dmst' in g.files:
            assert rel_err(mst.grad, g[b + 'dmst']) < 1e-4
        for k, v in blk.state_dict().items():
            if 'running' in k:
                assert rel_err(v, g[b + 'after.' + k]) < 1e-5, k


# ---------------------------------------------------------------------------- G2W / G3W: the bench kernels' widths
def wide_graph(V):
    g = GraphRef('ntu-rgb+d' if V == 25 else 'openpose', 'spatial_3')
    return tuple(torch.tensor(a, dtype=torch.float32) for a in (g.A, g.A2, g.A3))


def sub_err(key, got, g):
    """max |got - ref| / max(1, max|ref|) on the fixture's deterministic subsample of `got`, and the relative error of the
    full tensor's L2 norm (the fixture stores both; tests/golden/detinit.py)."""
    ref = torch.from_numpy(g[key]).double()
    got = got.detach().double().cpu()
    s = subsample(key, got, ref.numel())
    assert s.shape == ref.shape, (key, tuple(s.shape), tuple(ref.shape))
    norm, numel = g[key + '#norm']
    assert int(numel) == got.numel(), key
    e = float((s - ref).abs().max() / max(1.0, float(ref.abs().max())))
    en = abs(float(got.norm()) - float(norm)) / max(1.0, float(norm))
    return max(e, en)


@pytest.mark.parametrize('ci', range(len(WIDE_UNITS)))
def test_gcn_units_wide(ci, golden):
    """the seven GCN-unit variants at (64,64), (64,128), (128,256) channels (V = 25) and the plain unit at V = 18"""
    g = golden('units_g2w.npz')
    cin, cout, V = WIDE_UNITS[ci]
    A, A2, A3 = wide_graph(V)
    K = A.shape[0]
    x, r, W, bias = wide_unit_inputs(ci, K)
    b = 'w%d.' % ci
    for unit in wide_unit_names(ci):
        imps = [torch.from_numpy(g[b + 'imp%d' % j]).clone().requires_grad_(True) for j in (1, 2, 3)]
        kind = {'tgcn': 'plain', '3a': '3a', 'inc': 'incep', 'incnew': 'incep'}.get(unit, 'plain')
        u = R.RefGCN(kind, cin, cout, K)
        conv = u.branch.conv if kind == 'incep' else u.conv
        with torch.no_grad():
            conv.weight.copy_(W)
            conv.bias.copy_(bias)
        xx = x.clone().requires_grad_(True)
        if unit == 'tgcn':
            adj = (A * imps[0],)
        elif unit == '3a':
            adj = (A, imps[0], imps[1], imps[2])
        elif unit in ('inc', 'incnew'):
            adj = (A * imps[0], A2 * imps[1], A3 * imps[2])
        else:
            Ai = A * imps[0]
            adj = ({'multi3': Ai + Ai ** 2 + Ai ** 3, 'multi3fix': (Ai + Ai ** 2 + Ai ** 3) / 3, 'only3': Ai ** 3}[unit],)
        y = u(xx, adj)
        (y * r).sum().backward()
        k = b + unit
        assert sub_err(k + '.y', y, g) < 1e-5 and sub_err(k + '.dx', xx.grad, g) < 1e-5, unit
        assert sub_err(k + '.dW', conv.weight.grad, g) < 1e-5 and sub_err(k + '.db', conv.bias.grad, g) < 1e-5, unit
        for j in (1, 2, 3):
            if k + '.dimp%d' % j in g.files:
                assert sub_err(k + '.dimp%d' % j, imps[j - 1].grad, g) < 1e-5, (unit, j)


@pytest.mark.parametrize('kind', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn'])
def test_blocks_wide(kind, golden):
    """every st_gcn block variant at 64 -> 64 (stride 1) and 64 -> 128 (stride 2) channels: the widths of the trunk"""
    g = golden('block_g3w_%s.npz' % kind)
    seen = 0
    for si, (cin, cout, stride, V) in enumerate(WIDE_BLOCKS):
        if not wide_block_has(kind, si):
            continue
        seen += 1
        b = 'w%d.' % si
        A, A2, A3 = wide_graph(V)
        K = A.shape[0]
        x, r = wide_block_inputs(si, kind=kind)
        blk = R.RefBlock(kind, cin, cout, K, stride, dropout=0, residual=True)
        blk.load_state_dict(det_fill_(blk.state_dict(), salt=100 + si))
        imps = [torch.from_numpy(g[b + 'imp%d' % j]).clone().requires_grad_(True) for j in (1, 2, 3)]
        mst = torch.from_numpy(g[b + 'mst']).clone().requires_grad_(True)
        blk.eval()
        with torch.no_grad():
            assert sub_err(b + 'y_eval', blk(x, _block_adj(kind, A, A2, A3, imps), mst), g) < 1e-5
        blk.train()
        xx = x.clone().requires_grad_(True)
        y = blk(xx, _block_adj(kind, A, A2, A3, imps), mst)
        (y * r).sum().backward()
        assert sub_err(b + 'y_train', y, g) < 1e-5
        assert sub_err(b + 'dx', xx.grad, g) < 1e-4
        n_grad = 0
        for k, p in blk.named_parameters():
            if b + 'grad.' + k in g.files:
                assert sub_err(b + 'grad.' + k, p.grad, g) < 1e-4, k
                n_grad += 1
            else:
                assert p.grad is None, k
        assert n_grad > 0
        for j in (1, 2, 3):
            if b + 'dimp%d' % j in g.files:
                assert sub_err(b + 'dimp%d' % j, imps[j - 1].grad, g) < 1e-4
        if b + 'dmst' in g.files:
            assert rel_err(mst.grad, g[b + 'dmst']) < 1e-4
        for k, v in blk.state_dict().items():
            if 'running' in k:
                assert rel_err(v, g[b + 'after.' + k]) < 1e-5, k
    assert seen >= 2


@pytest.mark.parametrize('kind', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn'])
def test_wide_block_fixtures_have_no_knife_edge(kind, golden):
    """No ReLU pre-activation of a wide fixture case lies within fp32 round-off of zero (detinit.WIDE_X_SALT): the strict fp32
    gradient gates of the GPU test must not depend on which way such an element rounds."""
    g = golden('block_g3w_%s.npz' % kind)
    for si, (cin, cout, stride, V) in enumerate(WIDE_BLOCKS):
        if not wide_block_has(kind, si):
            continue
        b = 'w%d.' % si
        A, A2, A3 = wide_graph(V)
        x, _ = wide_block_inputs(si, kind=kind)
        blk = R.RefBlock(kind, cin, cout, A.shape[0], stride, dropout=0, residual=True)
        blk.load_state_dict(det_fill_(blk.state_dict(), salt=100 + si))
        imps = [torch.from_numpy(g[b + 'imp%d' % j]).clone() for j in (1, 2, 3)]
        mst = torch.from_numpy(g[b + 'mst']).clone()
        blk.train()
        rec = {}
        first = blk.tcn[0] if blk.tcn_kind == 'single' else blk.tcn_start[0]
        last = blk.tcn[4] if blk.tcn_kind == 'single' else blk.tcn_end[1]
        h1 = first.register_forward_hook(lambda m, i, o: rec.__setitem__('bn1', o.detach()))
        h2 = last.register_forward_hook(lambda m, i, o: rec.__setitem__('y', o.detach()))
        with torch.no_grad():
            res = 0 if blk.res_mode == 'none' else (x if blk.res_mode == 'id' else blk.residual(x))
            blk(x, _block_adj(kind, A, A2, A3, imps), mst)
        h1.remove(); h2.remove()
        m1, m2 = float(rec['bn1'].abs().min()), float((rec['y'] + res).abs().min())
        assert min(m1, m2) > 2e-6, (kind, si, m1, m2)


@pytest.mark.parametrize('tag', sorted(MODEL_CFG))
def test_state_dict_contract(tag):
    gargs, nc = MODEL_CFG[tag]
    m = R.RefModel(tag, 3, nc, gargs, True, dropout=0)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == SD[tag]
    assert [k for k, _ in m.named_parameters()] == SD[tag + '#param_names']
    assert sum(p.numel() for p in m.parameters()) == SD[tag + '#nparam']


@pytest.mark.parametrize('tag', sorted(MODEL_CFG))
def test_model_train_step(tag, golden):
    """G4b: one SGD-nesterov step of the harness counterpart (recognition.py:249-296)."""
    g = golden('model_g4_%s.npz' % tag)
    gargs, nc = MODEL_CFG[tag]
    m = R.RefModel(tag, 3, nc, gargs, True, dropout=0)
    m.load_state_dict(det_fill_(m.state_dict()))
    shp = tuple(int(s) for s in g['train_shape'])
    x = det_tensor('g4.xt.' + tag, shp)
    lab = det_labels('g4.lab.' + tag, shp[0], nc)
    assert np.array_equal(lab.numpy(), g['train_labels'])
    opt = R.make_optimizer(m)
    loss, logits = R.train_step(m, opt, x, lab)
    assert rel_err(logits, g['train_logits']) < 1e-4
    assert abs(float(loss) - float(g['train_loss'])) < 1e-4
    sd = m.state_dict()
    names = SD[tag + '#param_names']
    after = np.asarray([float(sd[k].double().norm()) for k in names])
    assert np.allclose(after, g['param_norms_after'], rtol=1e-4, atol=1e-6)
    for k in g.files:
        if k.startswith('after.'):
            assert rel_err(sd[k[6:]], g[k]) < 1e-4, k


@pytest.mark.parametrize('tag', ['st_gcnold', 'st_gcn_msgcn', 'st_gcn_mstcn_1x1', 'st_gcn_multi3_fix_3A_mstcn',
                                 'st_gcn_mstcn_1x1_deep'])
def test_model_eval_logits_full_clip(tag, golden):
    """G4a: eval-mode logits at the BASELINE clip shape (N=2, T=300/600, V=25/18, M=2)."""
    g = golden('model_g4_%s.npz' % tag)
    gargs, nc = MODEL_CFG[tag]
    m = R.RefModel(tag, 3, nc, gargs, True, dropout=0).eval()
    m.load_state_dict(det_fill_(m.state_dict()))
    x = det_tensor('g4.x.' + tag, tuple(int(s) for s in g['eval_shape']))
    with torch.no_grad():
        assert rel_err(m(x), g['eval_logits']) < 1e-4


@pytest.mark.parametrize('ci', range(2))
def test_mstcn_oracle_matches_reference(ci, golden):
    """G8: net/utils/ms_tcn.py:41-52 (BatchNorm -> ReLU -> conv_b -> the same BatchNorm -> Dropout), eval and train mode."""
    g = golden('mstcn_g8.npz')
    N, C, T, V, stride = [int(v) for v in g['c%d.shape' % ci]]
    m = R.RefMSTCN(C, 3, 9, 15, 0.0, stride=stride)
    m.load_state_dict(det_fill_(m.state_dict()))
    x = det_tensor('g8.x.%d' % ci, (N, C, T, V))
    with torch.no_grad():
        assert rel_err(m.eval()(x), g['c%d.eval' % ci]) < 1e-5
        assert rel_err(m.train()(x), g['c%d.train' % ci]) < 1e-5
    assert rel_err(m.batchnorm2d.running_mean, g['c%d.running_mean' % ci]) < 1e-5
    assert rel_err(m.batchnorm2d.running_var, g['c%d.running_var' % ci]) < 1e-5
    # gradients (round 5): autograd of the restatement against autograd of the reference, loss = sum(y * r), both modes
    for mode in ('train', 'eval'):
        m2 = R.RefMSTCN(C, 3, 9, 15, 0.0, stride=stride)
        m2.load_state_dict(det_fill_(m2.state_dict()))
        m2.train(mode == 'train')
        xg = x.clone().requires_grad_(True)
        y = m2(xg)
        (y * det_tensor('g8.r.%d' % ci, tuple(y.shape))).sum().backward()
        key = 'c%d.%s.' % (ci, mode)
        assert rel_err(xg.grad, g[key + 'dx']) < 1e-5 and rel_err(m2.conv_b.weight.grad, g[key + 'dW']) < 1e-5
        assert rel_err(m2.batchnorm2d.weight.grad, g[key + 'dgamma']) < 1e-5
        assert rel_err(m2.batchnorm2d.bias.grad, g[key + 'dbeta']) < 1e-5
        if mode == 'eval':
            assert rel_err(m2.conv_b.bias.grad, g[key + 'db']) < 1e-5


def test_model_full_clip_train_step_st_gcn_msgcn(golden):
    """G4L (round 5): one training step of BASELINE config 2's model at the FULL clip length (8 clips, T = 300): the oracle
    against the reference's logits, loss and every gradient norm.  (One model here -- ~15 s of CPU; the GPU suite checks all five.)"""
    tag = 'st_gcn_msgcn'
    g = golden('model_g4l_%s.npz' % tag)
    gargs, nc = MODEL_CFG[tag]
    m = R.RefModel(tag, 3, nc, gargs, True, dropout=0)
    m.load_state_dict(det_fill_(m.state_dict()))
    shp = tuple(int(v) for v in g['train_shape'])
    x = det_tensor('g4l.x.' + tag, shp)
    lab = det_labels('g4l.lab.' + tag, shp[0], nc)
    torch.set_num_threads(8)
    m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    loss.backward()
    assert rel_err(logits, g['train_logits']) < 1e-5 and abs(float(loss) - float(g['train_loss'])) < 1e-5
    gn = np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) for p in m.parameters()])
    assert np.array_equal(np.asarray([p.grad is None for p in m.parameters()]), g['grad_none'])
    assert np.allclose(gn, g['grad_norms'], rtol=2e-4, atol=1e-6)
