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


def l2rel(got, ref):
    """||got-ref|| / ||ref||: the bf16-storage metric (isolated ReLU-mask flips make max-norm meaningless there)."""
    got = got.detach().double().cpu()
    ref = torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape
    floor = 0.05 * ref.numel() ** 0.5      # RMS 0.05 per element: gradients that are ~0 by construction (conv bias
    return float((got - ref).norm() / max(floor, float(ref.norm())))   # in front of a train-mode BN) compare absolutely


def close(name, got, ref, tol, dt):
    """fp32: scale-aware max error < tol.  bf16 storage: relative L2 error < tol."""
    if dt == torch.float32:
        return diag(name, got, ref, tol) < tol
    refn = float(torch.as_tensor(ref).double().norm())
    if refn < 1e-3 * max(1, torch.as_tensor(ref).numel()) ** 0.5:
        # structurally-zero gradient (a conv bias in front of a train-mode BatchNorm): bf16 rounding of dz breaks the
        # exact cancellation; only require that it stays small
        return bool(got.detach().abs().max() < 0.5)
    ok = gate16(name + ' rel-L2 ' + str(dt)[6:], l2rel(got, ref), tol) and bool(torch.isfinite(got).all())
    if not ok:
        diag(name, got, ref, 0.0)
    return ok


def gate16(name, value, gate):
    """16-bit-storage error gates: the measured value goes to gpurun_out/err16_measured.txt next to its gate, so that the
    gates can be (and are) set to <= 2x what the kernels actually deliver -- a 2x regression must fail."""
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'err16_measured.txt'), 'a') as f:
        f.write('%-70s measured %.4g gate %.4g\n' % (name, value, gate))
    return value < gate


def sub_close(name, got, g, key, tol, dt, zero_gate=0.5):
    """`close()` against a WIDE fixture entry (tests/golden/detinit.py: a deterministic subsample of the reference tensor
    plus its L2 norm): fp32 -- scale-aware max error on the subsample and the norm; 16-bit storage -- relative L2 error on
    the subsample (recorded next to its gate in gpurun_out/err16_measured.txt)."""
    from detinit import subsample
    ref = torch.from_numpy(g[key]).double()
    got = got.detach().double().cpu()
    s = subsample(key, got, ref.numel())
    assert s.shape == ref.shape, (key, tuple(s.shape), tuple(ref.shape))
    norm, numel = g[key + '#norm']
    assert int(numel) == got.numel(), key
    if dt == torch.float32:
        e = float((s - ref).abs().max() / max(1.0, float(ref.abs().max())))
        en = abs(float(got.norm()) - float(norm)) / max(1.0, float(norm))
        ok = max(e, en) < tol and bool(torch.isfinite(got).all())
        if not ok:
            diag(name, s, ref, 0.0)
        return ok
    if float(ref.norm()) < 1e-3 * max(1, ref.numel()) ** 0.5:
        return gate16(name + ' max-abs (structural zero) ' + str(dt)[6:], float(got.abs().max()), zero_gate)   # (see close())
    ok = gate16(name + ' rel-L2 ' + str(dt)[6:], l2rel(s, ref), tol) and bool(torch.isfinite(got).all())
    if not ok:
        diag(name, s, ref, 0.0)
    return ok
