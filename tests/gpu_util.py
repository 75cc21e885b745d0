"""Helpers for the -m gpu parity tests (HIP product vs oracle / golden fixtures)."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')


def dev():
    return torch.device('cuda:0')


def to_ntvc(x):
    """reference layout (N, C, T, V) -> library layout (N, T, V, C)"""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nctv(x):
    return x.permute(0, 3, 1, 2).contiguous()


def diag(name, got, ref, tol):
    """relative error (scale-aware); on failure dump where the mismatch lives to gpurun_out/ for offline triage."""
    got = got.detach().double().cpu()
    ref = torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(ref.shape))
    scale = max(1.0, float(ref.abs().max()))
    err = (got - ref).abs() / scale
    worst = float(err.max()) if err.numel() else 0.0
    if not (worst < tol) or not torch.isfinite(got).all():
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, 'diag_%s.txt' % name.replace('/', '_')), 'w') as f:
            f.write('%s shape=%s worst=%g tol=%g scale=%g finite=%s\n' % (name, tuple(got.shape), worst, tol, scale,
                                                                      bool(torch.isfinite(got).all())))
            bad = (err >= tol) | ~torch.isfinite(got)
            f.write('bad fraction %g\n' % float(bad.double().mean()))
            idx = bad.nonzero()[:40]
            for i in idx:
                t = tuple(int(v) for v in i)
                f.write('%s got=%g ref=%g\n' % (t, float(got[t]), float(ref[t])))
            for d in range(got.dim()):
                other = [k for k in range(got.dim()) if k != d]
                prof = bad.double().mean(dim=other) if other else bad.double()
                f.write('bad-rate along dim %d: %s\n' % (d, ' '.join('%.2f' % float(v) for v in prof[:300])))
    return worst
