#!/usr/bin/env python3
"""Generate the golden fixtures by RUNNING THE REFERENCE (read-only, imported from
/root/reference).  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Writes small .npz/.json files next to this script.  Nothing from the reference other
than its *outputs* on seeded inputs is stored (G1..G5).  The GPU box has no
/root/reference: tests only read the committed fixtures.
"""
import itertools
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('ISTGCN_REFERENCE', '/root/reference')
sys.path.insert(0, HERE)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

from detinit import (det_fill_, det_tensor, det_labels, subsample, WIDE_UNITS, WIDE_BLOCKS, WIDE_T, wide_unit_inputs,  # noqa: E402
                     wide_unit_names, wide_block_inputs, wide_block_has)

torch.set_num_threads(8)

LAYOUTS = ['openpose', 'openpose_gravity', 'openpose_sym', 'ntu-rgb+d', 'ntu-rgb+d_half',
           'ntu-rgb+d_gravity', 'ntu-rgb+d_sym', 'ntu_edge']
STRATEGIES = ['uniform', 'distance', 'spatial', 'spatial_half', 'openpose_gravity',
              'ntu-rgb+d_gravity', 'spatial_3', 'spatial_sym', 'spatial_3_sym']


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print('%-34s %8.1f KB' % (name, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- G1 graph
def g1_graph():
    from net.utils.graph import Graph
    out, manifest = {}, {}
    for lay, st in itertools.product(LAYOUTS, STRATEGIES):
        tag = '%s|%s' % (lay, st)
        try:
            g = Graph(layout=lay, strategy=st)
        except Exception as e:  # noqa: BLE001 - the exception type IS the fixture
            manifest[tag] = {'error': type(e).__name__}
            continue
        rec = {'num_node': g.num_node, 'center': g.center, 'K': int(g.A.shape[0]), 'has_A23': hasattr(g, 'A2')}
        out[tag + '|A'] = g.A.astype(np.float64)
        out[tag + '|edge'] = np.asarray(g.edge, dtype=np.int64)
        if hasattr(g, 'A2'):
            out[tag + '|A2'] = g.A2.astype(np.float64)
            out[tag + '|A3'] = g.A3.astype(np.float64)
        manifest[tag] = rec
    # non-default ctor args used nowhere upstream but part of the signature
    g = Graph(layout='ntu-rgb+d', strategy='spatial', max_hop=2)
    out['ntu-rgb+d|spatial|max_hop2|A'] = g.A
    save('graph_g1.npz', **out)
    with open(os.path.join(HERE, 'graph_g1.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    ok = sum('error' not in v for v in manifest.values())
    print('G1: %d working layout x strategy pairs, %d failing' % (ok, len(manifest) - ok))


# ----------------------------------------------------------------------------- G2 units
def _A_for(V, strategy='spatial'):
    from net.utils.graph import Graph
    g = Graph(layout='ntu-rgb+d' if V == 25 else 'openpose', strategy=strategy)
    A = torch.tensor(g.A, dtype=torch.float32)
    extra = ()
    if hasattr(g, 'A2'):
        extra = (torch.tensor(g.A2, dtype=torch.float32), torch.tensor(g.A3, dtype=torch.float32))
    return (A,) + extra


def g2_units():
    import importlib
    out = {}
    cases = [(2, 3, 16, 12, 25), (2, 16, 16, 12, 25), (2, 16, 16, 12, 18), (1, 24, 40, 7, 25)]
    for ci, (n, cin, cout, T, V) in enumerate(cases):
        A, A2, A3 = _A_for(V, 'spatial_3')
        K = A.shape[0]
        x = det_tensor('g2.x.%d' % ci, (n, cin, T, V))
        r = det_tensor('g2.r.%d' % ci, (n, cout, T, V))
        W = det_tensor('g2.W.%d' % ci, (K * cout, cin, 1, 1), scale=cin ** -0.5)
        b = det_tensor('g2.b.%d' % ci, (K * cout,), scale=0.1)
        imps = [0.5 + torch.rand((K, V, V), generator=torch.Generator().manual_seed(100 + ci * 3 + j))
                for j in range(3)]
        base = 'c%d.' % ci
        out[base + 'x'], out[base + 'r'], out[base + 'W'], out[base + 'b'] = x, r, W, b
        out[base + 'A'], out[base + 'A2'], out[base + 'A3'] = A, A2, A3
        for j in range(3):
            out[base + 'imp%d' % (j + 1)] = imps[j]

        def run(unit_mod, cls, call):
            mod = importlib.import_module(unit_mod)
            u = getattr(mod, cls)(cin, cout, K)
            conv = u.conv if hasattr(u, 'conv') else u.branch.conv
            with torch.no_grad():
                conv.weight.copy_(W)
                conv.bias.copy_(b)
            xx = x.clone().requires_grad_(True)
            leaves = [t.clone().requires_grad_(True) for t in imps]
            y = call(u, xx, leaves)
            (y * r).sum().backward()
            return y, xx.grad, conv.weight.grad, conv.bias.grad, [t.grad for t in leaves]

        # a2  ConvTemporalGraphical: caller passes A*importance (st_gcnold.py:86)
        y, dx, dW, db, dimp = run('net.utils.tgcn', 'ConvTemporalGraphical',
                                  lambda u, xx, L: u(xx, A * L[0])[0])
        out.update({base + 'tgcn.y': y, base + 'tgcn.dx': dx, base + 'tgcn.dW': dW,
                    base + 'tgcn.db': db, base + 'tgcn.dimp1': dimp[0]})
        # a3  3A unit: raw A + three importances (tgcn_multi3_fix_3A.py:76-92)
        y, dx, dW, db, dimp = run('net.utils.tgcn_multi3_fix_3A', 'ConvTemporalGraphical',
                                  lambda u, xx, L: u(xx, A, L[0], L[1], L[2])[0])
        out.update({base + '3a.y': y, base + '3a.dx': dx, base + '3a.dW': dW, base + '3a.db': db})
        for j in range(3):
            out[base + '3a.dimp%d' % (j + 1)] = dimp[j]
        # a4  Inception2: caller pre-multiplies (st_gcn_msgcn.py:116-117)
        for tag, modname in (('inc', 'net.utils.inceptionv2_gcn'), ('incnew', 'net.utils.inceptionv2_gcn_new')):
            y, dx, dW, db, dimp = run(modname, 'Inception2',
                                      lambda u, xx, L: u(xx, A * L[0], A2 * L[1], A3 * L[2])[0])
            out.update({base + tag + '.y': y, base + tag + '.dx': dx, base + tag + '.dW': dW, base + tag + '.db': db})
            for j in range(3):
                out[base + tag + '.dimp%d' % (j + 1)] = dimp[j]
        # the "free" variants: same kernel, different folded adjacency
        for tag, modname in (('multi3', 'net.utils.tgcn_multi3'), ('multi3fix', 'net.utils.tgcn_multi3_fix'),
                             ('only3', 'net.utils.tgcn_only3')):
            y, dx, dW, db, dimp = run(modname, 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A * L[0])[0])
            out.update({base + tag + '.y': y, base + tag + '.dx': dx, base + tag + '.dimp1': dimp[0]})
    save('units_g2.npz', **out)


# ----------------------------------------------------------------------------- G3 blocks
BLOCK_KINDS = {
    # kind: (module, forward-arg builder)
    'st_gcnold': 'net.st_gcnold',
    'st_gcn_msgcn': 'net.st_gcn_msgcn',
    'st_gcn_mstcn': 'net.st_gcn_mstcn',
    'st_gcn_mstcn_1x1': 'net.st_gcn_mstcn_1x1',
    'st_gcn_multi3_fix_3A_mstcn': 'net.st_gcn_multi3_fix_3A_mstcn',
}
BLOCK_SHAPES = [(3, 16, 1, False), (16, 16, 1, True), (8, 16, 2, True)]


def block_args(kind, A, A2, A3, imps, mst):
    if kind == 'st_gcnold':
        return (A * imps[0],)
    if kind == 'st_gcn_msgcn':
        return (A * imps[0], A2 * imps[1], A3 * imps[2])
    if kind in ('st_gcn_mstcn', 'st_gcn_mstcn_1x1'):
        return (A * imps[0], mst)
    if kind == 'st_gcn_multi3_fix_3A_mstcn':
        return (A, imps[0], imps[1], imps[2], mst)
    raise KeyError(kind)


def g3_blocks():
    import importlib
    for kind, modname in BLOCK_KINDS.items():
        mod = importlib.import_module(modname)
        out = {}
        for V in (25, 18):
            A, A2, A3 = _A_for(V, 'spatial_3')
            K = A.shape[0]
            for si, (cin, cout, stride, residual) in enumerate(BLOCK_SHAPES):
                if V == 18 and si != 1:
                    continue
                n, T = 2, 16
                base = 'v%d.s%d.' % (V, si)
                blk = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=residual)
                sd = blk.state_dict()
                det_fill_(sd, salt=si + 10 * V)
                blk.load_state_dict(sd)
                x = det_tensor('g3.x' + base, (n, cin, T, V))
                r = det_tensor('g3.r' + base, (n, cout, T // stride, V))
                imps = [(0.5 + torch.rand((K, V, V), generator=torch.Generator().manual_seed(7 + j))).requires_grad_(True)
                        for j in range(3)]
                mst = (0.5 + torch.rand(3, generator=torch.Generator().manual_seed(11))).requires_grad_(True)
                for k, v in sd.items():
                    out[base + 'sd.' + k] = v.clone()
                out[base + 'x'], out[base + 'r'] = x, r
                out[base + 'A'], out[base + 'A2'], out[base + 'A3'] = A, A2, A3
                out[base + 'mst'] = mst.detach().clone()
                for j in range(3):
                    out[base + 'imp%d' % (j + 1)] = imps[j].detach().clone()
                # eval forward (running stats)
                blk.eval()
                with torch.no_grad():
                    out[base + 'y_eval'] = blk(x, *block_args(kind, A, A2, A3, imps, mst))[0]
                # train forward + backward (batch stats, dropout 0)
                blk.train()
                xx = x.clone().requires_grad_(True)
                y = blk(xx, *block_args(kind, A, A2, A3, imps, mst))[0]
                (y * r).sum().backward()
                out[base + 'y_train'] = y
                out[base + 'dx'] = xx.grad
                for k, p in blk.named_parameters():
                    if p.grad is not None:
                        out[base + 'grad.' + k] = p.grad
                for j in range(3):
                    if imps[j].grad is not None:
                        out[base + 'dimp%d' % (j + 1)] = imps[j].grad
                if mst.grad is not None:
                    out[base + 'dmst'] = mst.grad
                for k, v in blk.state_dict().items():
                    if 'running' in k:
                        out[base + 'after.' + k] = v.clone()
        save('block_g3_%s.npz' % kind, **out)



# ----------------------------------------------------------------------------- G2W / G3W: the bench kernels' widths
# Units and blocks at (C_in, C_out) = (64,64), (64,128), (128,256): the shapes the register-chained graph-conv kernels,
# the lean temporal conv and the bottleneck stream kernels serve (G2 / G3 above have 16-40 channels and never reach them).
# Inputs, upstream gradients and weights are regenerated from detinit by name (nothing but OUTPUTS is stored); tensors
# with more than 16384 elements are stored as a deterministic subsample (detinit.subsample_index) plus their L2 norm.
def _put(out, key, t, n=16384):
    out[key] = subsample(key, t, n).to(torch.float32)
    out[key + '#norm'] = np.asarray([float(t.detach().double().norm()), float(t.numel())])


def g2w_units():
    import importlib
    out = {}
    for ci, (cin, cout, V) in enumerate(WIDE_UNITS):
        A, A2, A3 = _A_for(V, 'spatial_3')
        K = A.shape[0]
        x, r, W, b = wide_unit_inputs(ci, K)
        imps = [0.5 + torch.rand((K, V, V), generator=torch.Generator().manual_seed(300 + ci * 3 + j)) for j in range(3)]
        base = 'w%d.' % ci
        for j in range(3):
            out[base + 'imp%d' % (j + 1)] = imps[j]

        def run(unit_mod, cls, call):
            mod = importlib.import_module(unit_mod)
            u = getattr(mod, cls)(cin, cout, K)
            conv = u.conv if hasattr(u, 'conv') else u.branch.conv
            with torch.no_grad():
                conv.weight.copy_(W)
                conv.bias.copy_(b)
            xx = x.clone().requires_grad_(True)
            leaves = [t.clone().requires_grad_(True) for t in imps]
            y = call(u, xx, leaves)
            (y * r).sum().backward()
            return y, xx.grad, conv.weight.grad, conv.bias.grad, [t.grad for t in leaves]

        units = [('tgcn', 'net.utils.tgcn', 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A * L[0])[0])]
        if V == 25:
            units += [('3a', 'net.utils.tgcn_multi3_fix_3A', 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A, L[0], L[1], L[2])[0]),
                      ('inc', 'net.utils.inceptionv2_gcn', 'Inception2', lambda u, xx, L: u(xx, A * L[0], A2 * L[1], A3 * L[2])[0]),
                      ('incnew', 'net.utils.inceptionv2_gcn_new', 'Inception2', lambda u, xx, L: u(xx, A * L[0], A2 * L[1], A3 * L[2])[0])]
        if ci == 0:
            units += [('multi3', 'net.utils.tgcn_multi3', 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A * L[0])[0]),
                      ('multi3fix', 'net.utils.tgcn_multi3_fix', 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A * L[0])[0]),
                      ('only3', 'net.utils.tgcn_only3', 'ConvTemporalGraphical', lambda u, xx, L: u(xx, A * L[0])[0])]
        assert [u_[0] for u_ in units] == wide_unit_names(ci)
        for tag, modname, cls, call in units:
            y, dx, dW, db, dimp = run(modname, cls, call)
            _put(out, base + tag + '.y', y)
            _put(out, base + tag + '.dx', dx)
            _put(out, base + tag + '.dW', dW, 8192)
            _put(out, base + tag + '.db', db)
            for j in range(3):
                if dimp[j] is not None:
                    _put(out, base + tag + '.dimp%d' % (j + 1), dimp[j])
    save('units_g2w.npz', **out)


def g3w_blocks(only=None):
    import importlib
    for kind, modname in BLOCK_KINDS.items():
        if only and kind not in only:
            continue                                       # (`g3w=kind[,kind]`: regenerate those files only)
        mod = importlib.import_module(modname)
        out = {}
        for si, (cin, cout, stride, V) in enumerate(WIDE_BLOCKS):
            if not wide_block_has(kind, si):
                continue                                   # (V = 18 is BASELINE config 3: the bottleneck model; + the plain block)
            A, A2, A3 = _A_for(V, 'spatial_3')
            K = A.shape[0]
            n, T = 2, WIDE_T
            base = 'w%d.' % si
            blk = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=True)
            sd = blk.state_dict()
            det_fill_(sd, salt=100 + si)
            blk.load_state_dict(sd)
            x, r = wide_block_inputs(si, kind=kind)
            imps = [(0.5 + torch.rand((K, V, V), generator=torch.Generator().manual_seed(17 + j))).requires_grad_(True)
                    for j in range(3)]
            mst = (0.5 + torch.rand(3, generator=torch.Generator().manual_seed(21))).requires_grad_(True)
            out[base + 'shape'] = np.asarray([cin, cout, stride, V, n, T])
            out[base + 'mst'] = mst.detach().clone()
            for j in range(3):
                out[base + 'imp%d' % (j + 1)] = imps[j].detach().clone()
            blk.eval()
            with torch.no_grad():
                _put(out, base + 'y_eval', blk(x, *block_args(kind, A, A2, A3, imps, mst))[0])
            blk.train()
            xx = x.clone().requires_grad_(True)
            y = blk(xx, *block_args(kind, A, A2, A3, imps, mst))[0]
            (y * r).sum().backward()
            _put(out, base + 'y_train', y)
            _put(out, base + 'dx', xx.grad)
            for k, p in blk.named_parameters():
                if p.grad is not None:
                    _put(out, base + 'grad.' + k, p.grad, 8192)
            for j in range(3):
                if imps[j].grad is not None:
                    _put(out, base + 'dimp%d' % (j + 1), imps[j].grad)
            if mst.grad is not None:
                out[base + 'dmst'] = mst.grad
            for k, v in blk.state_dict().items():
                if 'running' in k:
                    out[base + 'after.' + k] = v.clone()
        save('block_g3w_%s.npz' % kind, **out)

# ----------------------------------------------------------------------------- G4/G5 models
MODELS = {
    # tag: (module, graph_args, num_class, (N,T,V) for the eval fixture)
    'st_gcnold': ('net.st_gcnold', dict(layout='ntu-rgb+d', strategy='spatial'), 60, (2, 300, 25)),
    'st_gcn_msgcn': ('net.st_gcn_msgcn', dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, (2, 300, 25)),
    'st_gcn_mstcn_1x1': ('net.st_gcn_mstcn_1x1', dict(layout='openpose', strategy='spatial'), 400, (2, 300, 18)),
    'st_gcn_multi3_fix_3A_mstcn': ('net.st_gcn_multi3_fix_3A_mstcn', dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, (2, 300, 25)),
    'st_gcn_mstcn_1x1_deep': ('net.st_gcn_mstcn_1x1_deep', dict(layout='ntu-rgb+d', strategy='spatial'), 60, (2, 600, 25)),
    # not BASELINE configs, same kernels
    'st_gcn_mstcn': ('net.st_gcn_mstcn', dict(layout='ntu-rgb+d', strategy='spatial'), 60, (2, 300, 25)),
    'st_gcn_msgcn_new': ('net.st_gcn_msgcn_new', dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, (2, 300, 25)),
    'st_gcn_deep_msgcn': ('net.st_gcn_deep_msgcn', dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, (2, 300, 25)),
}
TRAIN_SHAPE = (4, 48)  # (N, T) of the one-SGD-step fixture


def g45_models():
    import importlib
    manifest = {}
    for tag, (modname, gargs, nc, (N, T, V)) in MODELS.items():
        mod = importlib.import_module(modname)
        torch.manual_seed(0)
        m = mod.Model(3, nc, gargs, True, dropout=0)
        sd = m.state_dict()
        manifest[tag] = {k: list(v.shape) for k, v in sd.items()}
        manifest[tag + '#nparam'] = int(sum(p.numel() for p in m.parameters()))
        det_fill_(sd)
        m.load_state_dict(sd)
        out = {}
        # G4a: eval-mode logits at the full clip shape
        x = det_tensor('g4.x.' + tag, (N, 3, T, V, 2))
        m.eval()
        with torch.no_grad():
            out['eval_logits'] = m(x)
        out['eval_shape'] = np.asarray([N, 3, T, V, 2])
        # G4b: one training step = recognition.py:249-296 (train mode, CE, SGD nesterov)
        n2, t2 = TRAIN_SHAPE
        xt = det_tensor('g4.xt.' + tag, (n2, 3, t2, V, 2))
        lab = det_labels('g4.lab.' + tag, n2, nc)
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
        logits = m(xt)
        loss = torch.nn.functional.cross_entropy(logits, lab)
        opt.zero_grad()
        loss.backward()
        names = [k for k, _ in m.named_parameters()]
        out['train_logits'] = logits
        out['train_loss'] = loss.detach()
        out['train_shape'] = np.asarray([n2, 3, t2, V, 2])
        out['train_labels'] = lab
        out['grad_norms'] = np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) for p in m.parameters()])
        out['grad_none'] = np.asarray([p.grad is None for p in m.parameters()])
        opt.step()
        after = m.state_dict()
        out['param_norms_after'] = np.asarray([float(after[k].double().norm()) for k in names])
        out['buffer_norms_after'] = np.asarray([float(v.double().norm()) for k, v in after.items()
                                                if 'running' in k])
        for k in ('fcn.bias', 'edge_importance.0', 'st_gcn_networks.0.gcn.conv.weight', 'data_bn.running_mean',
                  'st_gcn_networks.0.gcn.branch.conv.weight', 'mstcn_importance.1'):
            if k in after:
                out['after.' + k] = after[k]
        manifest[tag + '#param_names'] = names
        save('model_g4_%s.npz' % tag, **out)
    with open(os.path.join(HERE, 'state_dict_g5.json'), 'w') as f:
        json.dump(manifest, f, indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- G4L: a training step at the FULL clip length
# the whole-model tests at bench size compared the HIP path with itself; the reference-pinned ones were N = 2
# (eval) and (4, T = 48) (training).  G4L is one training step of each BASELINE model at its full clip length (T = 300; 600 for the
# deep model) on 8 clips (4 for T = 600): logits, loss, every gradient norm -- the NM = 16 sequences fill 16 x 300 x 25 = 120 000
# positions per layer, every kernel walks many tiles per workgroup.
G4L = {'st_gcnold': 8, 'st_gcn_msgcn': 8, 'st_gcn_mstcn_1x1': 8, 'st_gcn_multi3_fix_3A_mstcn': 8, 'st_gcn_mstcn_1x1_deep': 4}
# G4B: the SAME for BASELINE configs 2 and 4 at their bench shape -- 64 clips x (3, 300, 25, 2) -- so that the steps bench.py
# times have a reference twin (`make_golden.py g4b`: two minutes each; 25 GB of host memory for config 2, 45 GB for config 4,
# the full IST-GCN -- configs 3 and 5 at their 256 / 128 clips do not fit this container's 64 GB)
G4B = {'st_gcn_msgcn': 64, 'st_gcn_multi3_fix_3A_mstcn': 64}


def g4l_models(table=None, prefix='g4l'):
    import importlib
    for tag, n in (table or G4L).items():
        modname, gargs, nc, (_, T, V) = MODELS[tag]
        mod = importlib.import_module(modname)
        torch.manual_seed(0)
        m = mod.Model(3, nc, gargs, True, dropout=0)
        sd = m.state_dict()
        det_fill_(sd)
        m.load_state_dict(sd)
        xt = det_tensor(prefix + '.x.' + tag, (n, 3, T, V, 2))
        lab = det_labels(prefix + '.lab.' + tag, n, nc)
        m.train()
        logits = m(xt)
        loss = torch.nn.functional.cross_entropy(logits, lab)
        loss.backward()
        out = {'train_logits': logits, 'train_loss': loss.detach(), 'train_shape': np.asarray([n, 3, T, V, 2]),
               'grad_norms': np.asarray([0.0 if p.grad is None else float(p.grad.double().norm()) for p in m.parameters()]),
               'grad_none': np.asarray([p.grad is None for p in m.parameters()])}
        for k in ('fcn.weight', 'edge_importance.0', 'st_gcn_networks.9.gcn.conv.weight', 'st_gcn_networks.9.gcn.branch.conv.weight',
                  'data_bn.weight'):
            p = dict(m.named_parameters()).get(k)
            if p is not None and p.grad is not None:
                _put(out, 'grad.' + k, p.grad, 4096)
        save('model_%s_%s.npz' % (prefix, tag), **out)


# ----------------------------------------------------------------------------- G6 inference / extract_feature
def g6_extract_feature():
    """`extract_feature` (net/st_gcnold.py:98-120, caller processor/demo_offline.py:68-98) and the eval
    logits of the same clip.  Only st_gcnold: the msgcn / mstcn variants' extract_feature is broken upstream
    (st_gcn_msgcn.py:145-146 calls the block with too few arguments)."""
    import importlib
    tag = 'st_gcnold'
    modname, gargs, nc, _ = MODELS[tag]
    m = importlib.import_module(modname).Model(3, nc, gargs, True, dropout=0)
    sd = det_fill_(m.state_dict())
    m.load_state_dict(sd)
    m.eval()
    x = det_tensor('g6.x.' + tag, (1, 3, 32, 25, 2))
    with torch.no_grad():
        output, feature = m.extract_feature(x)
        logits = m(x)
    save('infer_g6_%s.npz' % tag, shape=np.asarray(x.shape), output=output, feature=feature, logits=logits)


# ----------------------------------------------------------------------------- G7 feeder augmentation
def g7_feeder_tools():
    """feeder/tools.py:31-101 (`auto_pading`, `random_choose`, `random_move`) on seeded clips.  The random
    draws are captured as explicit parameters by replaying the SAME generator calls the functions make, in the same
    order (random.randint for the offsets; random.choice then four np.random.choice for random_move), after re-seeding."""
    import random
    from feeder import tools
    out = {}
    C, T, V, M = 3, 40, 25, 2
    for case in range(4):
        d = det_tensor('g7.x.%d' % case, (C, T, V, M)).numpy().astype(np.float32)
        if case == 3:
            d[:, 30:] = 0                                        # trailing empty frames, as real NTU clips have
        out['c%d.x' % case] = d
        # auto_pading, begin = 0 (feeder.py:83-84: window_size > 0 without random_choose)
        out['c%d.pad64' % case] = tools.auto_pading(d.copy(), 64)
        # random_choose: crop (T > size) and random pad (T < size)
        for size in (24, 64):
            random.seed(100 + case)
            out['c%d.choose%d' % (case, size)] = tools.random_choose(d.copy(), size)
            random.seed(100 + case)
            out['c%d.choose%d.begin' % (case, size)] = np.asarray(
                random.randint(0, T - size) if T > size else random.randint(0, size - T))
        # random_move (default candidates, move_time 1)
        random.seed(200 + case)
        np.random.seed(200 + case)
        out['c%d.move' % case] = tools.random_move(d.copy())
        random.seed(200 + case)
        np.random.seed(200 + case)
        random.choice([1])
        A = np.random.choice([-10., -5., 0., 5., 10.], 2)
        S = np.random.choice([0.9, 1.0, 1.1], 2)
        Tx = np.random.choice([-0.2, -0.1, 0.0, 0.1, 0.2], 2)
        Ty = np.random.choice([-0.2, -0.1, 0.0, 0.1, 0.2], 2)
        out['c%d.move.nodes' % case] = np.stack([A, S, Tx, Ty])          # [4][2] node values
        # the training feeder's chain (feeder.py:81-86): random_choose then random_move
        random.seed(300 + case)
        np.random.seed(300 + case)
        out['c%d.chain24' % case] = tools.random_move(tools.random_choose(d.copy(), 24))
        random.seed(300 + case)
        np.random.seed(300 + case)
        out['c%d.chain24.begin' % case] = np.asarray(random.randint(0, T - 24))
        random.choice([1])
        A = np.random.choice([-10., -5., 0., 5., 10.], 2)
        S = np.random.choice([0.9, 1.0, 1.1], 2)
        Tx = np.random.choice([-0.2, -0.1, 0.0, 0.1, 0.2], 2)
        Ty = np.random.choice([-0.2, -0.1, 0.0, 0.1, 0.2], 2)
        out['c%d.chain24.nodes' % case] = np.stack([A, S, Tx, Ty])
    save('feeder_g7.npz', **out)


# ----------------------------------------------------------------------------- G8 MSTCN (dead upstream)
def g8_mstcn():
    """net/utils/ms_tcn.py:41-52: BatchNorm -> ReLU -> conv_b -> the SAME BatchNorm -> Dropout.  Train mode (dropout 0:
    batch statistics twice, running statistics updated twice) and eval mode, two shapes (stride 1 and 2)."""
    from net.utils.ms_tcn import MSTCN
    out = {}
    for ci, (N, C, T, V, stride) in enumerate([(2, 16, 12, 25, 1), (3, 32, 15, 18, 2)]):
        m = MSTCN(C, 3, 9, 15, 0.0, stride=stride)
        sd = det_fill_(m.state_dict())
        m.load_state_dict(sd)
        x = det_tensor('g8.x.%d' % ci, (N, C, T, V))
        imp = torch.ones(3)
        m.eval()
        with torch.no_grad():
            out['c%d.eval' % ci] = m(x.clone(), imp)
        m.train()
        with torch.no_grad():
            out['c%d.train' % ci] = m(x.clone(), imp)
        out['c%d.running_mean' % ci] = m.batchnorm2d.running_mean
        out['c%d.running_var' % ci] = m.batchnorm2d.running_var
        out['c%d.shape' % ci] = np.asarray([N, C, T, V, stride])
        # gradients (round 5; nothing upstream trains the class, autograd of its forward is the definition): loss =
        # sum(y * r) on a FRESH module per mode, so that the running statistics above stay those of two updates
        for mode in ('train', 'eval'):
            m2 = MSTCN(C, 3, 9, 15, 0.0, stride=stride)
            m2.load_state_dict(det_fill_(m2.state_dict()))
            m2.train(mode == 'train')
            xg = x.clone().requires_grad_(True)
            y = m2(xg * 1.0, imp)                  # (the reference's ReLU / Dropout are in-place: not on a leaf)
            r = det_tensor('g8.r.%d' % ci, tuple(y.shape))
            (y * r).sum().backward()
            out['c%d.%s.dx' % (ci, mode)] = xg.grad
            out['c%d.%s.dW' % (ci, mode)] = m2.conv_b.weight.grad
            out['c%d.%s.db' % (ci, mode)] = m2.conv_b.bias.grad
            out['c%d.%s.dgamma' % (ci, mode)] = m2.batchnorm2d.weight.grad
            out['c%d.%s.dbeta' % (ci, mode)] = m2.batchnorm2d.bias.grad
            assert m2.conv_a.weight.grad is None and m2.conv_c.weight.grad is None
    save('mstcn_g8.npz', **out)


if __name__ == '__main__':
    what = sys.argv[1:] or ['g1', 'g2', 'g3', 'g45', 'g6', 'g7', 'g8', 'g2w', 'g3w', 'g4l']
    if 'g1' in what:
        g1_graph()
    if 'g2' in what:
        g2_units()
    if 'g3' in what:
        g3_blocks()
    if 'g45' in what:
        g45_models()
    if 'g6' in what:
        g6_extract_feature()
    if 'g7' in what:
        g7_feeder_tools()
    if 'g8' in what:
        g8_mstcn()
    if 'g2w' in what:
        g2w_units()
    if 'g4l' in what:
        g4l_models()
    if 'g4b' in what:
        g4l_models(G4B, 'g4b')
    if 'g3w' in what:
        g3w_blocks()
    for w in what:
        if w.startswith('g3w='):
            g3w_blocks(w[4:].split(','))
