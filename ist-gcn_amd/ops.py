"""Thin PyTorch <-> C-ABI glue: every function here checks shapes/devices (raising before any
launch, like the reference's bare asserts, e.g. net/utils/tgcn.py:77), hands raw device
pointers and the current HIP stream of the tensor's device to `libistgcn_hip.so`, and raises
RuntimeError on a non-zero return.  Nothing here computes on the CPU; there is no fallback.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib

DT_F32, DT_BF16 = 0, 1
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}
STATS_REP = 8          # replicated BatchNorm partial-sum rows (spreads the fp64 atomics)


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError('istgcn: unsupported activation dtype %s (float32 / bfloat16 only)' % t.dtype)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check_dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('istgcn: the IST-GCN hot path runs on MI355X only (got a %s tensor); '
                               'there is no CPU fallback' % t.device)
        if not t.is_contiguous():
            raise RuntimeError('istgcn: non-contiguous tensor passed to a kernel')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('istgcn: tensors on different devices')
    return dev


def _call(fn_name, *args):
    lib = _lib.load()
    rc = getattr(lib, fn_name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed with code %d (%s)' % (fn_name, rc, {1: 'invalid argument', 2: 'launch failure'}.get(rc, '?')))


def gcn_geometry(cin, cout, K, dt):
    vals = [ctypes.c_int() for _ in range(5)]
    _call('istgcn_gcn_geometry', cin, cout, K, dt, *[ctypes.byref(v) for v in vals])
    return tuple(v.value for v in vals)  # CCeff, nch, KKp, MTtot, EPL


def pack_gcn_weight(wr, dtype):
    """wr: [Cout][K][Cin] fp32 (Wr[c][k][i]) -> fragment-ordered tensor the MFMA loop streams (see istgcn.h)."""
    cout, K, cin = wr.shape
    cce, nch, kkp, mttot, epl = gcn_geometry(cin, cout, K, _DT[dtype])
    w = F.pad(wr, (0, nch * cce - cin))
    w = w.reshape(cout, K, nch, cce).permute(2, 0, 1, 3).reshape(nch, cout, K * cce)
    w = F.pad(w, (0, kkp - K * cce, 0, mttot * 32 - cout))
    nkg = kkp // (2 * epl)
    w = w.reshape(nch, mttot, 32, nkg, 2, epl).permute(0, 1, 3, 4, 2, 5)
    return w.to(dtype).contiguous()


def gcn_forward(x, A, wp, cout, bterm=None, addend=None, out=None, stats=None, Tout=None, Tlog=None,
                in_t_stride=1, out_t_stride=1, nnz_cap=None, grid_cap=0):
    """istgcn_gcn_fwd.  x: [NM,Tin,V,Cin]; A: [K,V,V] fp32; wp from pack_gcn_weight; returns y [NM,Tout,V,cout]."""
    NM, Tin, V, Cin = x.shape
    K = A.shape[0]
    assert A.shape == (K, V, V) and A.dtype == torch.float32
    if Tlog is None:
        Tlog = (Tin - 1) // in_t_stride + 1
    if Tout is None:
        Tout = (Tlog - 1) * out_t_stride + 1
    if out is None:
        out = torch.empty((NM, Tout, V, cout), dtype=x.dtype, device=x.device)
    assert out.shape == (NM, Tout, V, cout) and out.dtype == x.dtype
    if addend is not None:
        assert addend.shape == out.shape and addend.dtype == x.dtype
    if bterm is not None:
        assert bterm.shape == (V, cout) and bterm.dtype == torch.float32
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.shape[-2:] == (2, cout)
    if nnz_cap is None:
        nnz_cap = K * V * V
    _check_dev(x, A, wp, bterm, addend, out, stats)
    _call('istgcn_gcn_fwd', _ptr(x), _ptr(A), _ptr(wp), _ptr(bterm), _ptr(addend), _ptr(out), _ptr(stats),
          0 if stats is None else stats.shape[0], None, NM, Tin, Tout, Tlog, V, Cin, cout, K,
          in_t_stride, out_t_stride, int(nnz_cap), dtype_code(x), grid_cap, _stream(x))
    return out
