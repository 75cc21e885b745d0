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


# ----------------------------------------------------------------------------------------------
# temporal convolution (istgcn_tconv)
# ----------------------------------------------------------------------------------------------
def _int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def tconv_geometry(V, cin, cout, tap_off, in_mul, dt):
    vals = [ctypes.c_int() for _ in range(4)]
    _call('istgcn_tconv_geometry', V, cin, cout, len(tap_off), _int_array(tap_off), in_mul, dt,
          *[ctypes.byref(v) for v in vals])
    return tuple(v.value for v in vals)  # CC, nch, MTtot, EPL


def pack_tconv_weight(wf, V, tap_off, in_mul, dtype):
    """wf: [ntaps][Cout][Cin] fp32 -> [nch][ntaps][MTtot][NKG][2][32][EPL] fragments (see istgcn.h)."""
    ntaps, cout, cin = wf.shape
    cc, nch, mttot, epl = tconv_geometry(V, cin, cout, tap_off, in_mul, _DT[dtype])
    nkg = cc // (2 * epl)
    w = F.pad(wf, (0, nch * cc - cin, 0, mttot * 32 - cout))
    w = w.reshape(ntaps, mttot, 32, nch, nkg, 2, epl).permute(3, 0, 1, 4, 5, 2, 6)
    return w.to(dtype).contiguous()


def tconv(x, wp, cout, tap_off, bias=None, pre=None, pre_relu=False, aux=None, maux=None, out=None, stats=None,
          mode=0, Tout=None, Mlog=None, in_mul=1, out_mul=1, out_off=0, grid_cap=0):
    """istgcn_tconv.  x: [NM,Tin,V,Cin] -> out [NM,Tout,V,cout]; see include/istgcn.h for the index algebra."""
    NM, Tin, V, Cin = x.shape
    assert Mlog is not None and Tout is not None
    if out is None:
        out = torch.empty((NM, Tout, V, cout), dtype=x.dtype, device=x.device)
    assert out.shape == (NM, Tout, V, cout) and out.dtype == x.dtype
    if bias is not None:
        assert bias.shape == (cout,) and bias.dtype == torch.float32
    if pre is not None:
        assert pre.shape == (2, Cin) and pre.dtype == torch.float32
    if mode == 1:
        assert aux is not None and aux.shape == out.shape and aux.dtype == x.dtype
        assert maux is not None and maux.shape == (4, cout) and maux.dtype == torch.float32
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.shape[-2:] == (2, cout)
    _check_dev(x, wp, bias, pre, aux, maux, out, stats)
    _call('istgcn_tconv', _ptr(x), _ptr(wp), _ptr(bias), _ptr(pre), int(bool(pre_relu)), _ptr(aux), _ptr(maux),
          _ptr(out), _ptr(stats), 0 if stats is None else stats.shape[0], mode, NM, Tin, Tout, Mlog, V, Cin, cout,
          len(tap_off), _int_array(tap_off), in_mul, out_mul, out_off, dtype_code(x), grid_cap, _stream(x))
    return out


def conv_taps_fwd(k, stride):
    """(tap offsets, in_mul) of a (k,1) Conv2d with padding (k-1)//2: in frame = stride*m + j - pad."""
    pad = (k - 1) // 2
    return [j - pad for j in range(k)], stride


def conv_taps_bwd(k, stride, phase):
    """taps of the data gradient that land on output frames t = stride*m + phase: [(j, d_j)], dz frame = m + d_j."""
    pad = (k - 1) // 2
    return [(j, (phase + pad - j) // stride) for j in range(k) if (phase + pad - j) % stride == 0]


def tconv_wgrad(dz, g, tap_off, in_mul=1, pre=None, pre_relu=False, want_bias=True, grid_cap=0):
    """istgcn_tconv_wgrad -> (dWf [ntaps][Cout][Cin] fp32, dbias [Cout] fp32 or None)."""
    NM, Tz, V, Cout = dz.shape
    NM2, Tin, V2, Cin = g.shape
    assert (NM, V) == (NM2, V2) and dz.dtype == g.dtype
    if pre is not None:
        assert pre.shape == (2, Cin) and pre.dtype == torch.float32
    dW = torch.zeros((len(tap_off), Cout, Cin), dtype=torch.float32, device=dz.device)
    db = torch.zeros((Cout,), dtype=torch.float32, device=dz.device) if want_bias else None
    _check_dev(dz, g, pre, dW, db)
    _call('istgcn_tconv_wgrad', _ptr(dz), _ptr(g), _ptr(pre), int(bool(pre_relu)), _ptr(dW), _ptr(db), NM, Tin, Tz,
          V, Cin, Cout, len(tap_off), _int_array(tap_off), in_mul, dtype_code(dz), grid_cap, _stream(dz))
    return dW, db
