"""Thin PyTorch <-> C-ABI glue: every function here checks shapes/devices (raising before any
launch, like the reference's bare asserts, e.g. net/utils/tgcn.py:77), hands raw device
pointers and the current HIP stream of the tensor's device to `libistgcn_hip.so`, and raises
RuntimeError on a non-zero return.  Nothing here computes on the CPU; there is no fallback.
"""
import ctypes
import os
import threading

import torch
import torch.nn.functional as F

from . import _lib

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}
STATS_REP = int(os.environ.get('ISTGCN_STATS_REP', '8'))          # replicated BatchNorm partial-sum rows (spreads the fp64 atomics)


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError('istgcn: unsupported activation dtype %s (float32 / bfloat16 / float16 only)' % t.dtype)


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


# Optional per-launch timing used by bench.py for the roofline line: when PROFILE is a list, every kernel launch
# appends (entry point, algorithmic flops, algorithmic bytes, start event, end event), the events recorded on the
# stream the kernel was launched on.  None (default) = no events, no overhead.
PROFILE = None
PROFILE_ONLY = None     # if set: only launches of this entry point are timed (keeps the event overhead off the others)


# Experiment hook: ISTGCN_GRID_CAPS="istgcn_bneck_in=512,istgcn_gcn_fwd=128" replaces a grid_cap of 0 (= as many workgroups as are
# resident) in the named entry points' wrappers (tools: sweeps of the persistent-grid sizes over a whole step).
GRID_CAPS = {k: int(v) for k, v in (kv.split('=') for kv in os.environ.get('ISTGCN_GRID_CAPS', '').split(',') if '=' in kv)}


def _gcap(name, grid_cap):
    return grid_cap if grid_cap else GRID_CAPS.get(name, 0)


def _call(fn_name, *args, work=None, dev=None, family=None):
    """dev: device of the tensors (from _check_dev).  The library launches on the CURRENT HIP device, so when the
    tensors live elsewhere (a model moved with .to('cuda:1') while cuda:0 is current) the call runs under a device
    guard; in the common case (same device) this costs one integer comparison."""
    lib = _lib.load()
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _call(fn_name, *args, work=work, family=family)
    if PROFILE is not None and work is not None and (PROFILE_ONLY is None or PROFILE_ONLY == (family or fn_name)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, fn_name)(*args)
        e1.record()
        PROFILE.append((family or fn_name, work[0], work[1], e0, e1))      # (family: the breakdown row of bench.py)
    else:
        rc = getattr(lib, fn_name)(*args)
    if rc != 0:
        why = {1: 'invalid argument', 2: 'launch failure'}.get(rc)
        if why is None:
            why = 'hipError %d %s' % (rc % 1000, 'from the launch' if rc < 2000 else 'from hipFuncSetAttribute')
        raise RuntimeError('%s failed with code %d (%s)' % (fn_name, rc, why))


def _trace_dump():
    lib = _lib.load()
    n = lib.istgcn_trace(2, None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    lib.istgcn_trace(2, buf, n + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        cnt, name = line.split('\t', 1)
        out[name] = int(cnt)
    return out


class trace:
    """`with ops.trace() as tr: ...; tr.kernels` -> {kernel symbol: launches} of every kernel the library launched inside
    the block (istgcn_trace, csrc/trace.hip: process-wide, test instrumentation; nests -- tests/conftest.py records every
    GPU test's kernels around the test's own blocks).  `tr.ran('gcn_rc_bwd_kernel')`: a launched symbol contains that text."""
    _depth = 0

    def __enter__(self):
        if trace._depth == 0:
            _lib.load().istgcn_trace(1, None, 0)
        trace._depth += 1
        self._base = _trace_dump()
        self.kernels = {}
        return self

    def __exit__(self, *exc):
        now = _trace_dump()
        self.kernels = {k: v - self._base.get(k, 0) for k, v in now.items() if v > self._base.get(k, 0)}
        trace._depth -= 1
        if trace._depth == 0:
            _lib.load().istgcn_trace(0, None, 0)
        return False

    def ran(self, text):
        return any(text in k for k in self.kernels)


def _esz(t):
    return t.element_size()


def gcn_geometry(cin, cout, K, dt):
    vals = [ctypes.c_int() for _ in range(5)]
    _call('istgcn_gcn_geometry', cin, cout, K, dt, *[ctypes.byref(v) for v in vals])
    return tuple(v.value for v in vals)  # CCeff, nch, KKp, MTtot, EPL


def _src_ok(w):
    if not w.is_cuda or w.dtype != torch.float32:
        raise RuntimeError('istgcn: weight packers take fp32 tensors on the GPU (got %s on %s)' % (w.dtype, w.device))


def pack_gcn_weight(wr, dtype):
    """wr: [Cout][K][Cin] fp32 view (any strides: read in place) -> fragment-ordered tensor the MFMA loop streams
    (istgcn_pack_gcn, one launch; layout in istgcn.h).  `pack_gcn_weight_ref` is the torch-op specification."""
    cout, K, cin = wr.shape
    if not wr.is_cuda:
        return pack_gcn_weight_ref(wr, dtype)            # host tensors: the specification itself (tests, tooling)
    _src_ok(wr)
    n = _lib.load().istgcn_pack_gcn_elems(cin, cout, K, _DT[dtype])
    out = torch.empty(int(n), dtype=dtype, device=wr.device)
    so, sk, si = wr.stride()
    _call('istgcn_pack_gcn', _ptr(wr), ctypes.c_longlong(so), ctypes.c_longlong(sk), ctypes.c_longlong(si), _ptr(out),
          cin, cout, K, _DT[dtype], _stream(wr), dev=wr.device)
    return out


def pack_gcn_weight_ref(wr, dtype):
    """Specification of istgcn_pack_gcn in torch ops: wr [Cout][K][Cin] -> [nch][MTtot][NKG][2][32][EPL]."""
    cout, K, cin = wr.shape
    cce, nch, kkp, mttot, epl = gcn_geometry(cin, cout, K, _DT[dtype])
    w = F.pad(wr, (0, nch * cce - cin))
    w = w.reshape(cout, K, nch, cce).permute(2, 0, 1, 3).reshape(nch, cout, K * cce)
    w = F.pad(w, (0, kkp - K * cce, 0, mttot * 32 - cout))
    nkg = kkp // (2 * epl)
    w = w.reshape(nch, mttot, 32, nkg, 2, epl).permute(0, 1, 3, 4, 2, 5)
    out = w.to(dtype).contiguous()
    if _lib.load().istgcn_gcn_rc_layout(cin, cout, K, _DT[dtype]):
        out = out.reshape(-1)
        if dtype == torch.float32:
            # float32 section (csrc/gcn_rc_f32.hip): [jt][k][q][s4][h][c][4] = Wr[32 jt + c][k][64 q + 32 h + 4 s4 + e]
            q = wr.reshape(cout // 32, 32, K, cin // 64, 2, 8, 4).permute(0, 2, 3, 5, 4, 1, 6)
        else:
            # register-chained section (csrc/gcn_rc.hip): [jt][k][s][h][c][8] = Wr[32 jt + c][k][16 s + 8 h + e]
            cpad = (cin + 15) // 16 * 16                   # (the 3-channel first layer: one zero-padded k-step)
            q = F.pad(wr, (0, cpad - cin)).reshape(cout // 32, 32, K, cpad // 16, 2, 8).permute(0, 2, 3, 4, 1, 5)
        out = torch.cat([out, q.to(dtype).contiguous().reshape(-1)])
    return out


# Weights epoch: the inference plans of net/_model.py (folded + fragment-packed weights, BatchNorm running statistics)
# are cached; tensor version counters cannot see writers that go through raw pointers (istgcn_sgd_step, the running-
# statistics update inside istgcn_bn_finalize) or that run no Python at all (hipGraph replays).  Every such writer bumps
# this process-wide counter and the plan cache keys on it (over-invalidation only costs a rebuild).
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


CHECK_NNZ = False        # debug / test switch: verify (with a host sync) that a reduced nnz_cap really covers nnz(A)


def gcn_rc_serves(cin, cout, K, V, dtype):
    """True when istgcn_gcn_fwd takes its register-chained kernel for this shape in 16-bit storage (csrc/gcn_fwd.hip)."""
    return dtype != torch.float32 and V <= 32 and bool(_lib.load().istgcn_gcn_rc_layout(cin, cout, K, _DT[dtype]))


def gcn_forward(x, A, wp, cout, bterm=None, addend=None, out=None, stats=None, Tout=None, Tlog=None,
                in_t_stride=1, out_t_stride=1, nnz_cap=None, grid_cap=0):
    """istgcn_gcn_fwd.  x: [NM,Tin,V,Cin]; A: [K,V,V] fp32; wp from pack_gcn_weight; returns y [NM,Tout,V,cout].
    nnz_cap (default K*V*V = cannot overflow) sizes the in-LDS column lists; a smaller cap is a promise that
    nnz(A) <= cap (the Models derive it from their adjacency buffers' pattern).  With ops.CHECK_NNZ the kernel's
    overflow flag is read back and a broken promise raises instead of silently truncating the lists."""
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
    status = torch.zeros(1, dtype=torch.int32, device=x.device) if (CHECK_NNZ and nnz_cap < K * V * V) else None
    dv = _check_dev(x, A, wp, bterm, addend, out, stats)
    _call('istgcn_gcn_fwd', _ptr(x), _ptr(A), _ptr(wp), _ptr(bterm), _ptr(addend), _ptr(out), _ptr(stats),
          0 if stats is None else stats.shape[0], _ptr(status), NM, Tin, Tout, Tlog, V, Cin, cout, K,
          in_t_stride, out_t_stride, int(nnz_cap), dtype_code(x), _gcap('istgcn_gcn_fwd', grid_cap), _stream(x),
          work=(2.0 * NM * Tlog * V * cout * K * Cin + 2.0 * NM * Tlog * V * V * K * cout,      # 1x1 conv + dense einsum
                float(NM * Tlog * V) * (Cin + cout * (2 if addend is not None else 1)) * _esz(x)), dev=dv)
    if status is not None and int(status.item()) != 0:
        raise RuntimeError('istgcn_gcn_fwd: the adjacency has more non-zeros than nnz_cap=%d (lists truncated)' % nnz_cap)
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


def pack_tconv_weight(wf, V, tap_off, in_mul, dtype, tap_sel=None):
    """wf: [taps][Cout][Cin] fp32 view (any strides) -> [nch][ntaps][NKG][MTtot][2][32][EPL] fragments
    (istgcn_pack_tconv, one launch).  tap_sel: which taps of wf feed the len(tap_off) packed taps (default: all, in
    order) -- the data gradient packs a per-phase subset of the transposed view without materialising it."""
    if tap_sel is None:
        tap_sel = list(range(wf.shape[0]))
    assert len(tap_sel) == len(tap_off) and max(tap_sel) < wf.shape[0]
    _, cout, cin = wf.shape
    if not wf.is_cuda:
        return pack_tconv_weight_ref(wf[list(tap_sel)], V, tap_off, in_mul, dtype)
    _src_ok(wf)
    offs = _int_array(tap_off)
    n = _lib.load().istgcn_pack_tconv_elems(V, cin, cout, len(tap_off), offs, in_mul, _DT[dtype])
    if n < 0:
        raise RuntimeError('istgcn_pack_tconv_elems: invalid geometry')
    out = torch.empty(int(n), dtype=dtype, device=wf.device)
    st, so, si = wf.stride()
    _call('istgcn_pack_tconv', _ptr(wf), ctypes.c_longlong(st), ctypes.c_longlong(so), ctypes.c_longlong(si),
          _int_array(tap_sel), _ptr(out), V, cin, cout, len(tap_off), offs, in_mul, _DT[dtype], _stream(wf), dev=wf.device)
    return out


def pack_tconv_weight_ref(wf, V, tap_off, in_mul, dtype):
    """Specification of istgcn_pack_tconv in torch ops: wf [ntaps][Cout][Cin] -> [nch][ntaps][NKG][MTtot][2][32][EPL]."""
    ntaps, cout, cin = wf.shape
    cc, nch, mttot, epl = tconv_geometry(V, cin, cout, tap_off, in_mul, _DT[dtype])
    nkg = cc // (2 * epl)
    w = F.pad(wf, (0, nch * cc - cin, 0, mttot * 32 - cout))
    w = w.reshape(ntaps, mttot, 32, nch, nkg, 2, epl).permute(3, 0, 4, 1, 5, 2, 6)
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
    if mode == 2:
        assert stats is None
        assert aux is None or (aux.shape == out.shape and aux.dtype == x.dtype)
        assert maux is None or (maux.shape[0] >= 2 and maux.shape[1] == cout and maux.dtype == torch.float32)
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.shape[-2:] == (2, cout)
    dv = _check_dev(x, wp, bias, pre, aux, maux, out, stats)
    _call('istgcn_tconv', _ptr(x), _ptr(wp), _ptr(bias), _ptr(pre), int(bool(pre_relu)), _ptr(aux), _ptr(maux),
          _ptr(out), _ptr(stats), 0 if stats is None else stats.shape[0], mode, NM, Tin, Tout, Mlog, V, Cin, cout,
          len(tap_off), _int_array(tap_off), in_mul, out_mul, out_off, dtype_code(x), _gcap('istgcn_tconv', grid_cap), _stream(x),
          work=(2.0 * NM * Mlog * V * cout * Cin * len(tap_off),
                float(NM * V) * (min(Tin, Mlog * in_mul) * Cin + Mlog * cout * (2 if (mode == 1 or aux is not None) else 1)) * _esz(x)), dev=dv)
    return out


BNECK_RC = os.environ.get('ISTGCN_BNECK_RC', '1') != '0'     # A/B switch: False = the generic temporal-conv kernels
BNECK_FUSE_BN = os.environ.get('ISTGCN_BNECK_FUSE_BN', '1') != '0'   # A/B switch: False = affine2 writes dz as a tensor


def bneck_ok(V, C, Wn, Wp, dtype):
    """Do the register-chained bottleneck kernels serve this shape (istgcn_bneck_ok)?"""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(_lib.load().istgcn_bneck_ok(int(V), int(C), int(Wn), int(Wp), 1 if dtype == torch.bfloat16 else 2))


def bneck_in(x, W, Wp, bias=None, pre=None, pre_relu=False, grid_cap=0):
    """istgcn_bneck_in: x [..., C] -> [..., Wp] with y[..., n] = sum_c W[n, c] pre(x[..., c]) + bias[n] (n < W.shape[0], zeros
    above).  W: fp32 [Wn, C], ANY strides (a transposed view is read in place)."""
    C = x.shape[-1]
    Wn = W.shape[0]
    assert x.is_contiguous() and W.shape == (Wn, C) and W.dtype == torch.float32
    if bias is not None:
        assert bias.shape == (Wn,) and bias.dtype == torch.float32 and bias.is_contiguous()
    if pre is not None:
        assert pre.shape == (2, C) and pre.dtype == torch.float32 and pre.is_contiguous()
    rows = x.numel() // C
    y = torch.empty(x.shape[:-1] + (Wp,), dtype=x.dtype, device=x.device)
    dv = _check_dev(x, bias, pre, y)
    assert W.device == x.device                              # (strided view: not for _check_dev's contiguity test)
    _call('istgcn_bneck_in', _ptr(x), _ptr(W), ctypes.c_longlong(W.stride(0)), ctypes.c_longlong(W.stride(1)), _ptr(bias),
          _ptr(pre), int(bool(pre_relu)), _ptr(y), ctypes.c_longlong(rows), C, Wn, Wp, dtype_code(x), _gcap('istgcn_bneck_in', grid_cap), _stream(x),
          work=(2.0 * rows * C * Wn, float(rows) * (C + Wp) * _esz(x)), dev=dv, family='istgcn_bneck')
    return y


def bneck_out(q, Wt, tap_sel, off0, We, C, bt=None, be=None, aux=None, maux=None, stats=None, mode=0, Tout=None, Mlog=None,
              in_mul=1, out_mul=1, out_off=0, Wn=None, yb=None, z=None, grid_cap=0):
    """istgcn_bneck_out: q [NM,Tin,V,Wp] -> (yb [NM,Tout,V,Wp], z [NM,Tout,V,C]); include/istgcn.h has the index algebra.
    Wt: fp32 [k, r, c] weight slices (any strides; tap j of the launch = slice tap_sel[j], its input frame in_mul*m + off0 + j),
    We: fp32 [C, Wn] (any strides)."""
    NM, Tin, V, Wp = q.shape
    assert q.is_contiguous() and Wt.dim() == 3 and Wt.dtype == torch.float32 and We.dtype == torch.float32
    Wn = We.shape[1] if Wn is None else Wn
    assert We.shape == (C, Wn) and Wt.shape[1] == Wn and Wt.shape[2] == Wn
    assert Mlog is not None and Tout is not None
    if yb is None:
        yb = torch.empty((NM, Tout, V, Wp), dtype=q.dtype, device=q.device)
    if z is None:
        z = torch.empty((NM, Tout, V, C), dtype=q.dtype, device=q.device)
    assert yb.shape == (NM, Tout, V, Wp) and z.shape == (NM, Tout, V, C) and yb.is_contiguous() and z.is_contiguous()
    if bt is not None:
        assert bt.shape == (Wn,) and bt.dtype == torch.float32 and bt.is_contiguous()
    if be is not None:
        assert be.shape == (C,) and be.dtype == torch.float32 and be.is_contiguous()
    if mode == 1:
        assert aux is not None and aux.shape == z.shape and aux.dtype == q.dtype and aux.is_contiguous()
        assert maux is not None and maux.shape == (4, C) and maux.dtype == torch.float32 and maux.is_contiguous()
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.shape[-2:] == (2, C)
    dv = _check_dev(q, bt, be, aux, maux, stats, yb, z)
    assert Wt.device == q.device and We.device == q.device   # (strided views: not for _check_dev's contiguity test)
    _call('istgcn_bneck_out', _ptr(q), _ptr(Wt), ctypes.c_longlong(Wt.stride(0)), ctypes.c_longlong(Wt.stride(1)),
          ctypes.c_longlong(Wt.stride(2)), _int_array(tap_sel), len(tap_sel), int(off0), _ptr(bt), _ptr(yb), _ptr(We),
          ctypes.c_longlong(We.stride(0)), ctypes.c_longlong(We.stride(1)), _ptr(be), _ptr(z), _ptr(aux), _ptr(maux),
          _ptr(stats), 0 if stats is None else stats.shape[0], mode, NM, Tin, Tout, Mlog, V, C, Wn, Wp, in_mul, out_mul,
          out_off, dtype_code(q), _gcap('istgcn_bneck_out', grid_cap), _stream(q),
          work=(2.0 * NM * Mlog * V * Wn * (Wn * len(tap_sel) + C),
                float(NM * V) * (min(Tin, Mlog * in_mul) * Wp + Mlog * (Wp + C * (2 if mode == 1 else 1))) * _esz(q)),
          dev=dv, family='istgcn_bneck')
    return yb, z


def bneck_wgrad(wide, nrw, wide_is_out, pre=None, pre_relu=False, want_bias=True, grid_cap=0):
    """istgcn_bneck_wgrad -> (dW fp32 [C][Wp] (wide_is_out) or [Wp][C], db fp32 [C] (wide_is_out) or [Wp], or None):
    dW(n, c) = sum_p nrw[p][n] pre(wide[p][c]); the bias gradient is the column sum of the conv's OUTPUT gradient, i.e. of
    `wide` when the wide tensor is the output side (conv_1x1_end) and of `nrw` otherwise (conv_1x1_start)."""
    C, Wp = wide.shape[-1], nrw.shape[-1]
    assert wide.is_contiguous() and nrw.is_contiguous() and wide.dtype == nrw.dtype
    rows = wide.numel() // C
    assert nrw.numel() // Wp == rows
    if pre is not None:
        assert pre.shape == (2, C) and pre.dtype == torch.float32 and pre.is_contiguous()
    dW = torch.zeros((C, Wp) if wide_is_out else (Wp, C), dtype=torch.float32, device=wide.device)
    db = torch.zeros((C if wide_is_out else Wp,), dtype=torch.float32, device=wide.device) if want_bias else None
    dv = _check_dev(wide, nrw, pre, dW, db)
    _call('istgcn_bneck_wgrad', _ptr(wide), _ptr(nrw), _ptr(pre), int(bool(pre_relu)), _ptr(dW), _ptr(db),
          int(bool(wide_is_out)), int(bool(wide_is_out)), ctypes.c_longlong(rows), C, Wp, dtype_code(wide), _gcap('istgcn_bneck_wgrad', grid_cap),
          _ptr(_wgrad_ws(wide.device)), ctypes.c_longlong(WGRAD_WS_FLOATS), _stream(wide),
          work=(2.0 * rows * C * Wp, float(rows) * (C + Wp) * _esz(wide)), dev=dv, family='istgcn_bneck_wgrad')
    return dW, db


def bneck_wgrad_taps(dy, q, ntaps, off0, in_mul=1, want_bias=True, grid_cap=0):
    """istgcn_bneck_wgrad_taps -> (dW fp32 [ntaps][Wp][Wp], db fp32 [Wp] or None): dW[j][n'][n] = sum dy[m][v][n'] q[in_mul m + off0 + j][v][n]."""
    NM, Tz, V, Wp = dy.shape
    NM2, Tin, V2, Wp2 = q.shape
    assert (NM, V, Wp) == (NM2, V2, Wp2) and dy.dtype == q.dtype and dy.is_contiguous() and q.is_contiguous()
    dW = torch.zeros((ntaps, Wp, Wp), dtype=torch.float32, device=dy.device)
    db = torch.zeros((Wp,), dtype=torch.float32, device=dy.device) if want_bias else None
    dv = _check_dev(dy, q, dW, db)
    _call('istgcn_bneck_wgrad_taps', _ptr(dy), _ptr(q), _ptr(dW), _ptr(db), NM, Tin, Tz, V, Wp, ntaps, int(off0), in_mul,
          dtype_code(dy), _gcap('istgcn_bneck_wgrad_taps', grid_cap), _ptr(_wgrad_ws(dy.device)), ctypes.c_longlong(WGRAD_WS_FLOATS), _stream(dy),
          work=(2.0 * NM * Tz * V * Wp * Wp * ntaps, float(NM * V) * (Tz + Tin) * Wp * _esz(dy)), dev=dv,
          family='istgcn_bneck_wgrad')
    return dW, db


def bneck_bwd_in_ok(C, Wn, Wp, dtype):
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(_lib.load().istgcn_bneck_bwd_in_ok(int(C), int(Wn), int(Wp), 1 if dtype == torch.bfloat16 else 2))


def bneck_bwd_in(dres, z, abc, yb, W, Wp, p_drop=0.0, seed=0, epoch=None, grid_cap=0):
    """istgcn_bneck_bwd_in -> (dyb [..., Wp], dW fp32 [C][Wp], db fp32 [C]): dz = abc[0]*dropmask*dres + abc[1]*z + abc[2] formed
    in registers (never stored), dyb = W dz with W fp32 [Wn, C] (any strides: pass We.t()), dW = dz^T yb, db = sum dz."""
    C = dres.shape[-1]
    Wn = W.shape[0]
    assert dres.is_contiguous() and z.is_contiguous() and yb.is_contiguous() and dres.shape == z.shape and dres.dtype == z.dtype == yb.dtype
    assert abc.shape == (3, C) and abc.dtype == torch.float32 and abc.is_contiguous()
    assert W.shape == (Wn, C) and W.dtype == torch.float32 and yb.shape == dres.shape[:-1] + (Wp,)
    rows = dres.numel() // C
    dyb = torch.empty(dres.shape[:-1] + (Wp,), dtype=dres.dtype, device=dres.device)
    dW = torch.zeros((C, Wp), dtype=torch.float32, device=dres.device)
    db = torch.zeros((C,), dtype=torch.float32, device=dres.device)
    dv = _check_dev(dres, z, abc, yb, dyb, dW, db)
    assert W.device == dres.device
    _call('istgcn_bneck_bwd_in', _ptr(dres), _ptr(z), _ptr(abc), ctypes.c_float(p_drop), ctypes.c_ulonglong(seed),
          _epoch_ptr(epoch, dres), _ptr(yb), _ptr(W), ctypes.c_longlong(W.stride(0)), ctypes.c_longlong(W.stride(1)),
          _ptr(dyb), _ptr(dW), _ptr(db), ctypes.c_longlong(rows), C, Wn, Wp, dtype_code(dres), _gcap('istgcn_bneck_bwd_in', grid_cap),
          _ptr(_wgrad_ws(dres.device)), ctypes.c_longlong(WGRAD_WS_FLOATS), _stream(dres),
          work=(4.0 * rows * C * Wp + 4.0 * rows * C, float(rows) * (2 * C + 2 * Wp) * _esz(dres)), dev=dv, family='istgcn_bneck')
    return dyb, dW, db


def conv_taps_fwd(k, stride):
    """(tap offsets, in_mul) of a (k,1) Conv2d with padding (k-1)//2: in frame = stride*m + j - pad."""
    pad = (k - 1) // 2
    return [j - pad for j in range(k)], stride


def conv_taps_bwd(k, stride, phase):
    """taps of the data gradient that land on output frames t = stride*m + phase: [(j, d_j)], dz frame = m + d_j."""
    pad = (k - 1) // 2
    return [(j, (phase + pad - j) // stride) for j in range(k) if (phase + pad - j) % stride == 0]


class StepArena:
    """One zero-filled fp32 allocation per training step for the gradient accumulators of ALL blocks (one fill launch
    instead of one per block).  The size is the total the previous step asked for (`hint`); a request that does not fit
    (first step, changed shapes) gets its own allocation."""

    def __init__(self, hint=0):
        self.hint, self.requested, self.buf, self.off = int(hint), 0, None, 0

    def take(self, n, device):
        self.requested += n
        if self.buf is None and self.hint >= n:
            self.buf = torch.zeros(self.hint, dtype=torch.float32, device=device)
        if self.buf is not None and self.buf.device == device and self.off + n <= self.buf.numel():
            v = self.buf[self.off:self.off + n]
            self.off += n
            return v
        return torch.zeros(n, dtype=torch.float32, device=device)


class ZeroArena:
    """One zero-filled fp32 allocation carved into the gradient buffers of a block's backward (one memset launch instead
    of one per buffer; none at all when a StepArena `parent` provides the memory).  `take()` hands out consecutive
    16-byte aligned views."""

    def __init__(self, shapes, device, parent=None):
        self.sizes = [(int(torch.Size(sh).numel()) + 3) // 4 * 4 for sh in shapes]
        total = sum(self.sizes)
        self.buf = parent.take(total, device) if parent is not None else torch.zeros(total, dtype=torch.float32, device=device)
        self.shapes, self.i, self.off = list(shapes), 0, 0

    def take(self):
        sh, n = self.shapes[self.i], self.sizes[self.i]
        v = self.buf[self.off:self.off + int(torch.Size(sh).numel())].view(sh)
        self.i += 1
        self.off += n
        return v


_WGRAD_WS = {}
WGRAD_WS_FLOATS = 16 << 20          # 64 MiB per device: per-workgroup partial sums of the weight-gradient kernels


def _wgrad_ws(dev):
    """Partial-sum workspace of the wgrad kernels: one per (device, stream).  Launches on one stream are serialised (a
    launch writes its slices and its reduce kernel consumes them before the next launch starts), so host threads
    that share a stream may share it; launches that can overlap (different streams) never do."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(WGRAD_WS_FLOATS, dtype=torch.float32, device=dev)
    return ws


def tconv_wgrad(dz, g, tap_off, in_mul=1, pre=None, pre_relu=False, want_bias=True, grid_cap=0, out=None):
    """istgcn_tconv_wgrad -> (dWf [ntaps][Cout][Cin] fp32, dbias [Cout] fp32 or None)."""
    NM, Tz, V, Cout = dz.shape
    NM2, Tin, V2, Cin = g.shape
    assert (NM, V) == (NM2, V2) and dz.dtype == g.dtype
    if pre is not None:
        assert pre.shape == (2, Cin) and pre.dtype == torch.float32
    if out is not None:                 # caller-provided ZERO-filled (dW [ntaps][Cout][Cin], db [Cout] or None)
        dW, db = out
        assert dW.shape == (len(tap_off), Cout, Cin) and dW.dtype == torch.float32 and dW.is_contiguous()
    else:
        dW = torch.zeros((len(tap_off), Cout, Cin), dtype=torch.float32, device=dz.device)
        db = torch.zeros((Cout,), dtype=torch.float32, device=dz.device) if want_bias else None
    dv = _check_dev(dz, g, pre, dW, db)
    # want_bias=False with a caller-provided buffer: the column sums are skipped, `db` stays as the caller zero-filled it
    _call('istgcn_tconv_wgrad', _ptr(dz), _ptr(g), _ptr(pre), int(bool(pre_relu)), _ptr(dW), _ptr(db if want_bias else None), NM, Tin, Tz,
          V, Cin, Cout, len(tap_off), _int_array(tap_off), in_mul, dtype_code(dz), _gcap('istgcn_tconv_wgrad', grid_cap),
          _ptr(_wgrad_ws(dz.device)), ctypes.c_longlong(WGRAD_WS_FLOATS), _stream(dz),
          work=(2.0 * NM * Tz * V * Cout * Cin * len(tap_off), float(NM * V) * (Tz * Cout + Tin * Cin) * _esz(dz)), dev=dv)
    return dW, db


# ----------------------------------------------------------------------------------------------
# graph-conv backward: istgcn_gcn_bwd_data (dx + dA) and istgcn_gcn_wgrad (dW + S)
# ----------------------------------------------------------------------------------------------
def gcn_bwd_geometry(cin, cout, K, dt):
    vals = [ctypes.c_int() for _ in range(6)]
    _call('istgcn_gcn_bwd_geometry', cin, cout, K, dt, *[ctypes.byref(v) for v in vals])
    return tuple(v.value for v in vals)  # CCi, nchi, CCc, nchc, KKp, EPL


def pack_gcn_wb(w3, dtype):
    """w3: [K][Cout][Cin] fp32 view (any strides) -> fragments [nchi][nchc][NKGc][MTK][2][32][EPL] of the dxa product
    (istgcn_pack_gcn_bwd, one launch)."""
    K, cout, cin = w3.shape
    if not w3.is_cuda:
        return pack_gcn_wb_ref(w3, dtype)
    _src_ok(w3)
    n = _lib.load().istgcn_pack_gcn_bwd_elems(cin, cout, K, _DT[dtype])
    out = torch.empty(int(n), dtype=dtype, device=w3.device)
    sk, sc, si = w3.stride()
    _call('istgcn_pack_gcn_bwd', _ptr(w3), ctypes.c_longlong(sk), ctypes.c_longlong(sc), ctypes.c_longlong(si),
          _ptr(out), cin, cout, K, _DT[dtype], _stream(w3), dev=w3.device)
    return out


def pack_gcn_wb_ref(w3, dtype):
    """Specification of istgcn_pack_gcn_bwd in torch ops: w3 [K][Cout][Cin] -> [nchi][nchc][NKGc][MTK][2][32][EPL]:
    row kk = k*CCi + (i - ich*CCi) (zero padded to KKp), contraction index c = cch*CCc + kg*2*EPL + h*EPL + e."""
    K, cout, cin = w3.shape
    cci, nchi, ccc, nchc, kkp, epl = gcn_bwd_geometry(cin, cout, K, _DT[dtype])
    w = F.pad(w3.permute(2, 0, 1), (0, nchc * ccc - cout, 0, 0, 0, nchi * cci - cin))      # [Cin_p][K][Cout_p]
    w = w.reshape(nchi, cci, K, nchc, ccc).permute(0, 2, 1, 3, 4).reshape(nchi, K * cci, nchc, ccc)
    w = F.pad(w, (0, 0, 0, 0, 0, kkp - K * cci))
    nkg = ccc // (2 * epl)
    w = w.reshape(nchi, kkp // 32, 32, nchc, nkg, 2, epl).permute(0, 3, 4, 1, 5, 2, 6)
    out = w.to(dtype).contiguous()
    if _lib.load().istgcn_gcn_bwd_rc_layout(cin, cout, K, _DT[dtype]):
        # register-chained section (csrc/gcn_rc_bwd.hip): [it][k][s][h][c][8] = W3[k][16 s + 8 h + e][32 it + p(c)],
        # p(c) = c with bits 2 and 3 swapped
        cs = torch.arange(32)
        pc = (cs & ~12) | ((cs & 4) << 1) | ((cs & 8) >> 1)
        cpad = (cin + 31) // 32 * 32                       # (the 3-channel first layer: one zero-padded tile)
        q = F.pad(w3, (0, cpad - cin)).reshape(K, cout // 16, 2, 8, cpad // 32, 32)[..., pc]            # [k][s][h][e][it][c]
        q = q.permute(4, 0, 1, 2, 5, 3)                                        # [it][k][s][h][c][e]
        out = torch.cat([out.reshape(-1), q.to(dtype).contiguous().reshape(-1)])
    return out


def first_layer_da_only(cin, cout, K, dtype, V):
    """Can istgcn_gcn_bwd_data skip dx (dx = NULL) for this shape?  The models' 3-channel first layer in 16-bit storage: its
    input needs no gradient, only the adjacency gradient does (csrc/gcn_rc_bwd.hip, narrow variant)."""
    import os
    if os.environ.get('ISTGCN_GCN_RC', '1') == '0':        # (A/B switch of the library: round-2 kernels only)
        return False
    return cin == 3 and V <= 32 and dtype != torch.float32 and bool(_lib.load().istgcn_gcn_bwd_rc_layout(cin, cout, K, _DT[dtype]))


def gcn_bwd_addend_mask_ok(V, cin, cout, K, dtype):
    """Does istgcn_gcn_bwd_data take its addend as (tensor, ReLU byte mask) for this shape (the register-chained kernel)?"""
    return dtype != torch.float32 and bool(_lib.load().istgcn_gcn_bwd_addend_mask_ok(V, cin, cout, K, _DT[dtype]))


def gcn_bwd_data(dy, A, w3, x=None, addend=None, want_dA=True, nnz_cap=None, grid_cap=0, dA_out=None, pattern=None, wb=None,
                 want_dx=True, addend_mask=None):
    """istgcn_gcn_bwd_data -> (dx [NM,T,V,Cin], dA [K,V,V] fp32 or None).  pattern [K,V,V] fp32 (non-zero = entry whose
    gradient is wanted; None = the non-zeros of A): pass the constant adjacency of A = B * importance so that an
    importance value of exactly 0 keeps its gradient, or ones for a dense learnable A (autograd of tgcn.py:86)."""
    NM, T, V, Cout = dy.shape
    K, cout2, Cin = w3.shape
    assert cout2 == Cout and A.shape == (K, V, V) and A.dtype == torch.float32
    dev = dy.device
    if not want_dx and not (want_dA and addend is None and first_layer_da_only(Cin, Cout, K, dy.dtype, V)):
        want_dx = True                   # the dx-less form exists for the 16-bit first layer only
    dx = torch.empty((NM, T, V, Cin), dtype=dy.dtype, device=dev) if want_dx else None
    dA = None
    if want_dA:
        assert x is not None and x.shape == (NM, T, V, Cin) and x.dtype == dy.dtype
        dA = dA_out if dA_out is not None else torch.zeros((K, V, V), dtype=torch.float32, device=dev)
        assert dA.shape == (K, V, V) and dA.dtype == torch.float32 and dA.is_contiguous()
    if addend is not None:
        assert addend.shape == (NM, T, V, Cin) and addend.dtype == dy.dtype
    if wb is None:
        wb = pack_gcn_wb(w3, dy.dtype)
    if nnz_cap is None:
        nnz_cap = K * V * V
    if pattern is not None:
        assert pattern.shape == (K, V, V) and pattern.dtype == torch.float32
    if want_dA and V > 32 and nnz_cap > 4096:
        # the round-2 kernel keeps one slot per pattern entry in LDS (16 per thread of a 256-thread workgroup); the
        # register-chained kernel (V <= 32) has no such limit
        raise RuntimeError('istgcn_gcn_bwd_data: the adjacency gradient of a graph with V = %d > 32 joints supports at most '
                           '4096 pattern entries (got nnz_cap = %d): pass a sparser `pattern`' % (V, nnz_cap))
    if addend_mask is not None:
        assert addend is not None and addend_mask.dtype == torch.uint8 and addend_mask.numel() * 8 == addend.numel()
    dv = _check_dev(dy, x, A, pattern, wb, addend, addend_mask, dx, dA)
    _call('istgcn_gcn_bwd_data', _ptr(dy), _ptr(x if want_dA else None), _ptr(A), _ptr(pattern), _ptr(wb), _ptr(addend),
          _ptr(addend_mask), _ptr(dx), _ptr(dA), NM, T, V, Cin, Cout, K, int(nnz_cap), dtype_code(dy),
          _gcap('istgcn_gcn_bwd_data', grid_cap), _stream(dy),
          work=(2.0 * NM * T * V * Cout * K * Cin + 2.0 * NM * T * V * V * K * Cin,
                float(NM * T * V) * (Cout + Cin * (1 + (1 if want_dA else 0) + (1 if addend is not None else 0))) * _esz(dy)), dev=dv)
    return dx, dA


def gcn_wgrad(dy, x, A, want_S=True, nnz_cap=None, grid_cap=0, out=None):
    """istgcn_gcn_wgrad -> (dW [K][Cout][Cin] fp32, S [V][Cout] fp32 or None)."""
    NM, T, V, Cout = dy.shape
    Cin = x.shape[3]
    K = A.shape[0]
    assert x.shape[:3] == dy.shape[:3] and x.dtype == dy.dtype and A.shape == (K, V, V)
    dev = dy.device
    if out is not None:                 # caller-provided ZERO-filled (dW [K][Cout][Cin], S [V][Cout] or None)
        dW, S = out
        assert dW.shape == (K, Cout, Cin) and dW.dtype == torch.float32 and dW.is_contiguous()
    else:
        dW = torch.zeros((K, Cout, Cin), dtype=torch.float32, device=dev)
        S = torch.zeros((V, Cout), dtype=torch.float32, device=dev) if want_S else None
    if nnz_cap is None:
        nnz_cap = K * V * V
    dv = _check_dev(dy, x, A, dW, S)
    _call('istgcn_gcn_wgrad', _ptr(dy), _ptr(x), _ptr(A), _ptr(dW), _ptr(S), NM, T, V, Cin, Cout, K, int(nnz_cap),
          dtype_code(dy), _gcap('istgcn_gcn_wgrad', grid_cap), _ptr(_wgrad_ws(dev)), ctypes.c_longlong(WGRAD_WS_FLOATS), _stream(dy),
          work=(2.0 * NM * T * V * Cout * K * Cin, float(NM * T * V) * (Cout + Cin) * _esz(dy)), dev=dv)
    return dW, S


# ----------------------------------------------------------------------------------------------
# BatchNorm bookkeeping + the block's HBM-bound glue (pointwise.hip)
# ----------------------------------------------------------------------------------------------
def new_stats(C, device):
    return torch.zeros((STATS_REP, 2, C), dtype=torch.float64, device=device)


_STATS_SCRATCH = {}          # (device, stream, host thread, slot, C) -> [buffer, handed-out flag]
_SCRATCH_BY_PTR = {}         # data_ptr -> the same entry (so the consumer finds it without scanning)
_SCRATCH_LOCK = threading.Lock()


def stats_scratch(slot, C, device):
    """A persistent, zero-between-uses [STATS_REP][2][C] fp64 buffer: the kernel that produces batch sums adds into it,
    `bn_finalize` / `bn_bwd_coef` (clear=True) read it and zero it again.  One per (device, stream, host thread, slot,
    C): distinct `slot`s for sums that are alive at the same time, and the thread in the key so that two modules driven
    from two host threads onto one stream (or nn.DataParallel's replica threads) never add into each other's sums.
    Saves a memset launch per BatchNorm per pass.  If a previous user never consumed its sums (an exception between
    producer and consumer), the buffer is re-zeroed here."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), slot, C)
    ent = _STATS_SCRATCH.get(key)
    if ent is None:
        ent = [new_stats(C, device), False]
        with _SCRATCH_LOCK:
            _STATS_SCRATCH[key] = ent
            _SCRATCH_BY_PTR[ent[0].data_ptr()] = ent
    elif ent[1]:
        ent[0].zero_()
    ent[1] = True                      # handed to a producer; bn_finalize / bn_bwd_coef(clear=True) mark it clean
    return ent[0]


def _scratch_consumed(stats):
    ent = _SCRATCH_BY_PTR.get(stats.data_ptr())
    if ent is not None and ent[0] is stats:
        ent[1] = False


# "Last workgroup finalises" tails (csrc/bn_tail.hpp) instead of the stand-alone bn_finalize / bn_bwd_coef launches: built,
# tested (tests/test_gpu_block.py::test_bn_tails_equal_standalone_launches) and measured NEUTRAL (config 2: 14.49 vs 14.53
# ms/step, config 1: 7.46 vs 7.33, config 5: 46.9 vs 47.0 -- the serial tail of the last workgroup costs what the 5 us launch
# did), hence off by default; ISTGCN_BN_TAILS=1 turns them on.
BN_TAILS = os.environ.get('ISTGCN_BN_TAILS', '0') == '1'
# Round 5: the st_gcn block's backward does not write dres = dout * [out > 0]; its readers take dout + the ReLU byte mask
# (functional.STGCNBlockFn.backward).  ISTGCN_DRES_FREE=0: the tensor is written as before (A/B).
DRES_FREE = os.environ.get('ISTGCN_DRES_FREE', '1') != '0'
_TAIL = threading.local()
_TICKETS = {}                # stats data_ptr -> int32[1] ticket of the "last workgroup finalises" protocol (zero between launches)


def _ticket(stats):
    t = _TICKETS.get(stats.data_ptr())
    if t is None or t.device != stats.device:
        t = _TICKETS[stats.data_ptr()] = torch.zeros(1, dtype=torch.int32, device=stats.device)
    return t


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, training, clear=False, defer=False):
    """-> coef [4][C] fp32: scale, shift, mean, rstd.  Training also updates the running statistics in place.
    defer=True (training, clear=True; call it BEFORE launching the kernel that adds into `stats`): the arithmetic is armed
    as the tail of that kernel -- its last workgroup finalises (csrc/bn_tail.hpp) -- and `bn_tail_flush()` after the
    producer's launch runs the stand-alone kernel instead if the producer's variant has no tail."""
    C = gamma.shape[0]
    coef = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    dv = _check_dev(stats, gamma, beta, running_mean, running_var, coef)
    args = (stats, count, gamma, beta, running_mean, running_var, momentum, eps, training, clear, coef, dv)
    if defer:
        # (always computed AFTER the producer's launch: by its last workgroup when armed, by bn_tail_flush() otherwise)
        _drop_stale_tail()
        armed = BN_TAILS and training and clear and stats is not None
        if armed:
            _call('istgcn_bn_tail_arm_finalize', _ptr(stats), stats.shape[0], ctypes.c_double(float(count)), _ptr(gamma),
                  _ptr(beta), _ptr(running_mean), _ptr(running_var), ctypes.c_float(momentum), ctypes.c_float(eps), _ptr(coef),
                  C, _ptr(_ticket(stats)), dev=dv)
        _TAIL.pending = ('fin', args, armed)
        return coef
    _bn_finalize_now(*args)
    return coef


def _bn_finalize_now(stats, count, gamma, beta, running_mean, running_var, momentum, eps, training, clear, coef, dv):
    C = gamma.shape[0]
    _call('istgcn_bn_finalize', _ptr(stats), 0 if stats is None else stats.shape[0], int(bool(clear)),
          ctypes.c_double(float(count)),
          _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), ctypes.c_float(momentum),
          ctypes.c_float(eps), int(bool(training)), _ptr(coef), C, _stream(gamma), dev=dv)
    if clear and stats is not None:
        _scratch_consumed(stats)
    if training:
        bump_weights_epoch()              # running_mean / running_var were written through raw pointers


def bn_bwd_coef(stats, count, gamma, coef, training, clear=False, defer=False):
    """-> (abc [3][C], dgamma [C], dbeta [C]) from the two BatchNorm-backward sums.  defer: as in bn_finalize."""
    C = gamma.shape[0]
    abc = torch.empty((3, C), dtype=torch.float32, device=gamma.device)
    dg = torch.empty((C,), dtype=torch.float32, device=gamma.device)
    db = torch.empty((C,), dtype=torch.float32, device=gamma.device)
    dv = _check_dev(stats, gamma, coef, abc, dg, db)
    args = (stats, count, gamma, coef, training, clear, abc, dg, db, dv)
    if defer:
        _drop_stale_tail()
        armed = BN_TAILS and clear
        if armed:
            _call('istgcn_bn_tail_arm_bwd', _ptr(stats), stats.shape[0], ctypes.c_double(float(count)), _ptr(gamma), _ptr(coef),
                  int(bool(training)), _ptr(abc), _ptr(dg), _ptr(db), C, _ptr(_ticket(stats)), dev=dv)
        _TAIL.pending = ('bwd', args, armed)
        return abc, dg, db
    _bn_bwd_coef_now(*args)
    return abc, dg, db


def _bn_bwd_coef_now(stats, count, gamma, coef, training, clear, abc, dg, db, dv):
    C = gamma.shape[0]
    _call('istgcn_bn_bwd_coef', _ptr(stats), stats.shape[0], int(bool(clear)), ctypes.c_double(float(count)),
          _ptr(gamma), _ptr(coef),
          int(bool(training)), _ptr(abc), _ptr(dg), _ptr(db), C, _stream(gamma), dev=dv)
    if clear:
        _scratch_consumed(stats)


TAIL_STATS = {'taken': 0, 'standalone': 0}       # diagnostics: tails carried by a producer / run as their own launch


def _drop_stale_tail():
    # an exception between arming and flushing (the producer's launch raised) leaves a tail behind: forget it
    if getattr(_TAIL, 'pending', None) is not None:
        if _TAIL.pending[2]:
            _lib.load().istgcn_bn_tail_disarm()
        _TAIL.pending = None


def bn_tail_flush():
    """After the launch(es) of the producer of an armed tail: if no kernel took the tail (a variant without one), run the
    stand-alone kernel now; either way the deferred outputs are valid (in stream order) afterwards."""
    pend = getattr(_TAIL, 'pending', None)
    if pend is None:
        return
    _TAIL.pending = None
    kind, args, armed = pend
    if not armed or _lib.load().istgcn_bn_tail_disarm():
        TAIL_STATS['standalone'] += 1
        (_bn_finalize_now if kind == 'fin' else _bn_bwd_coef_now)(*args)
        return
    TAIL_STATS['taken'] += 1
    _scratch_consumed(args[0])                      # the tail zeroed the sums
    if kind == 'fin':
        bump_weights_epoch()


def bn_tail_armed():
    return getattr(_TAIL, 'pending', None) is not None


def _rows(t):
    return t.numel() // t.shape[-1]


def _epoch_ptr(epoch, like):
    """Device-resident dropout seed offset (int64[1] on the tensors' device) or None -> pointer argument."""
    if epoch is None:
        return None
    assert epoch.dtype == torch.int64 and epoch.numel() == 1 and epoch.device == like.device
    return _ptr(epoch)


def relu_mask_ok(C, dtype):
    """Is the one-byte-per-vector ReLU mask available for this channel count / storage type (istgcn_relu_mask_ok)?"""
    epl = 4 if dtype == torch.float32 else 8
    return C % epl == 0 and C // epl <= 256 and 256 % (C // epl) == 0


def block_out_fwd(z, coef2, res=None, coefr=None, p_drop=0.0, seed=0, epoch=None, want_mask=False):
    """-> out, or (out, relu_mask) with want_mask: relu_mask uint8 [rows*C/vector width] (None where relu_mask_ok is
    False) lets block_out_bwd skip its read of `out`."""
    out = torch.empty_like(z)
    C = z.shape[-1]
    rmask = None
    if want_mask and relu_mask_ok(C, z.dtype):
        rmask = torch.empty(z.numel() // (4 if z.dtype == torch.float32 else 8), dtype=torch.uint8, device=z.device)
    dv = _check_dev(z, coef2, res, coefr, out, rmask)
    if res is not None:
        assert res.shape == z.shape and res.dtype == z.dtype
    _call('istgcn_block_out_fwd', _ptr(z), _ptr(coef2), _ptr(res), _ptr(coefr), _ptr(out), _ptr(rmask),
          ctypes.c_longlong(_rows(z)), C, ctypes.c_float(p_drop), ctypes.c_ulonglong(seed), _epoch_ptr(epoch, z),
          dtype_code(z), _stream(z),
          work=(3.0 * z.numel(), float(z.numel()) * (3 if res is not None else 2) * _esz(z)), dev=dv)
    return (out, rmask) if want_mask else out


def block_out_bwd(dout, out, z, coef2, r=None, coefr=None, p_drop=0.0, seed=0, scratch=False, epoch=None, relu_mask=None,
                  tail=None, want_dres=True):
    """-> (dres = dout*[out>0], stats2, statsr or None); scratch=True: the sums go to `stats_scratch` slots 0 / 1
    (consume them with bn_bwd_coef(clear=True)).  relu_mask: the forward's byte mask; `out` is then not read.
    tail = (count, gamma, training): tcn.3's backward coefficients are computed by the kernel's last workgroup (or by the
    stand-alone kernel right behind it) and returned as a fourth value (abc, dgamma, dbeta)."""
    C = z.shape[-1]
    # want_dres=False (needs relu_mask): only the sums -- the consumers take dout and the byte mask themselves (affine2 with
    # relu_mask, gcn_bwd_data with addend_mask); the first returned value is then None
    assert want_dres or relu_mask is not None
    dres = torch.empty_like(z) if want_dres else None
    if scratch:
        st2 = stats_scratch(0, C, z.device)
        str_ = stats_scratch(1, C, z.device) if r is not None else None
    else:
        st2 = new_stats(C, z.device)
        str_ = new_stats(C, z.device) if r is not None else None
    assert dout.shape == z.shape and dout.dtype == z.dtype and (out is None or out.shape == z.shape)
    assert out is not None or relu_mask is not None
    dv = _check_dev(dout, out, relu_mask, z, coef2, r, coefr, dres, st2, str_)
    nt = (4 if r is not None else 3) + (1.0 / 16 if relu_mask is not None else 1) - (0 if want_dres else 1)
    res2 = None
    if tail is not None:
        res2 = bn_bwd_coef(st2, tail[0], tail[1], coef2, tail[2], clear=True, defer=True)
    _call('istgcn_block_out_bwd', _ptr(dout), _ptr(None if relu_mask is not None else out), _ptr(relu_mask), _ptr(z),
          _ptr(coef2), _ptr(r), _ptr(coefr), _ptr(dres), _ptr(st2), _ptr(str_), STATS_REP, ctypes.c_longlong(_rows(z)), C,
          ctypes.c_float(p_drop), ctypes.c_ulonglong(seed), _epoch_ptr(epoch, z), dtype_code(z), _stream(z),
          work=(6.0 * z.numel(), float(z.numel()) * nt * _esz(z)), dev=dv)
    if tail is not None:
        bn_tail_flush()
        return dres, st2, str_, res2
    return dres, st2, str_


def affine2(d, x, abc, p_drop=0.0, seed=0, epoch=None, relu_mask=None):
    """out = abc[0]*d*dropmask + abc[1]*x + abc[2]  (BatchNorm backward, elementwise part).  relu_mask: d := d * [bit] first
    (the forward's byte mask of block_out_fwd: d is then dout itself, not the dres block_out_bwd would have written)."""
    out = torch.empty_like(d)
    dv = _check_dev(d, relu_mask, x, abc, out)
    _call('istgcn_affine2m', _ptr(d), _ptr(relu_mask), _ptr(x), _ptr(abc), _ptr(out), ctypes.c_longlong(_rows(d)), d.shape[-1],
          ctypes.c_float(p_drop), ctypes.c_ulonglong(seed), _epoch_ptr(epoch, d), dtype_code(d), _stream(d),
          work=(4.0 * d.numel(), float(d.numel()) * ((3 if x is not None else 2) + (1.0 / 16 if relu_mask is not None else 0)) * _esz(d)),
          dev=dv, family='istgcn_affine2')
    return out


POOL_SLICES = 8


def pool_fwd(y, M):
    """Global average pooling (st_gcnold.py:89-91): y [N*M, T, V, C] -> feat [N, C] fp32 = mean over (T, V) and the M persons.
    One kernel writes partial sums over POOL_SLICES row slices per sequence; one torch reduction adds a clip's M*S rows."""
    NM, T, V, C = y.shape
    assert y.is_contiguous() and NM % M == 0
    S = min(POOL_SLICES, T * V)
    psum = torch.empty((NM, S, C), dtype=torch.float32, device=y.device)
    dv = _check_dev(y, psum)
    _call('istgcn_pool_fwd', _ptr(y), _ptr(psum), NM, T * V, C, S, dtype_code(y), _stream(y),
          work=(float(y.numel()), float(y.numel()) * _esz(y)), dev=dv)
    return psum.view(NM // M, M * S, C).sum(1) * (1.0 / (M * T * V))


def pool_bwd(dfeat, shape, dtype, M):
    """dy [N*M, T, V, C] in `dtype` = dfeat[n] / (M*T*V) at every position (backward of pool_fwd)."""
    NM, T, V, C = shape
    dfeat = dfeat.contiguous()
    assert dfeat.shape == (NM // M, C) and dfeat.dtype == torch.float32
    dy = torch.empty(shape, dtype=dtype, device=dfeat.device)
    dv = _check_dev(dfeat, dy)
    _call('istgcn_pool_bwd', _ptr(dfeat), _ptr(dy), NM, T * V, C, M, ctypes.c_float(1.0 / (M * T * V)), dtype_code(dy),
          _stream(dy), work=(float(dy.numel()), float(dy.numel()) * _esz(dy)), dev=dv)
    return dy


# ----------------------------------------------------------------------------------------------
# parameter folds of the graph-conv unit (fold.hip)
# ----------------------------------------------------------------------------------------------
def fold_fwd(B, imps, bias, C):
    """-> (A_eff [K,V,V], bterm [V,C] or None).  B: [J,K,V,V] constants, imps: J tensors [K,V,V], bias [K*C] or None."""
    J, K, V, _ = B.shape
    assert len(imps) == J and all(i.shape == (K, V, V) and i.dtype == torch.float32 for i in imps)
    A_eff = torch.empty((K, V, V), dtype=torch.float32, device=B.device)
    bterm = torch.empty((V, C), dtype=torch.float32, device=B.device) if bias is not None else None
    dv = _check_dev(B, bias, A_eff, bterm, *imps)
    ip = [_ptr(i) for i in imps] + [_ptr(None)] * (3 - J)
    _call('istgcn_fold_fwd', _ptr(B), J, ip[0], ip[1], ip[2], _ptr(bias), _ptr(A_eff), _ptr(bterm), K, V, int(C),
          _stream(B), dev=dv)
    return A_eff, bterm


def fold_bwd(B, imps, bias, dA, S, C):
    """-> ([dimp_j], dbias or None) from dA (grad of A_eff, or None) and S (grad of bterm, or None)."""
    J, K, V, _ = B.shape
    dimps = [torch.empty_like(i) for i in imps]
    dbias = torch.empty_like(bias) if (bias is not None and S is not None) else None
    if dbias is None and bias is not None:
        dbias = torch.zeros_like(bias)
    dv = _check_dev(B, bias, dA, S, dbias, *imps, *dimps)
    ip = [_ptr(i) for i in imps] + [_ptr(None)] * (3 - J)
    dp = [_ptr(i) for i in dimps] + [_ptr(None)] * (3 - J)
    _call('istgcn_fold_bwd', _ptr(B), J, ip[0], ip[1], ip[2], _ptr(bias), _ptr(dA), _ptr(S), dp[0], dp[1], dp[2],
          _ptr(dbias if S is not None else None), K, V, int(C), _stream(B), dev=dv)
    return dimps, dbias


def _ptr_table(ts):
    return (ctypes.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])


def fold_fwd_batch(B, imps, biases, Cs):
    """All blocks' folds in one launch.  B [J,K,V,V]; imps: nb lists of J tensors; biases: nb tensors or Nones; Cs: nb ints.
    -> ([A_eff_i] views of one [nb,K,V,V] buffer, [bterm_i or None])."""
    J, K, V, _ = B.shape
    nb = len(imps)
    A_all = torch.empty((nb, K, V, V), dtype=torch.float32, device=B.device)
    bts = [None if b is None else torch.empty((V, C), dtype=torch.float32, device=B.device) for b, C in zip(biases, Cs)]
    flat = [t for imp in imps for t in imp]
    dv = _check_dev(B, A_all, *flat, *[b for b in biases if b is not None])
    itab = _ptr_table([imp[j] if j < J else None for imp in imps for j in range(3)])
    _call('istgcn_fold_fwd_batch', nb, _ptr(B), J, itab, _ptr_table(biases), _ptr_table([A_all[i] for i in range(nb)]),
          _ptr_table(bts), _int_array(Cs), K, V, _stream(B), dev=dv)
    return [A_all[i] for i in range(nb)], bts


def fold_bwd_batch(B, imps, biases, dAs, Ss, Cs):
    """-> ([[dimp_ij]], [dbias_i or None]) for all blocks in one launch (dAs / Ss entries may be None)."""
    J, K, V, _ = B.shape
    nb = len(imps)
    dimp_all = torch.empty((nb, J, K, V, V), dtype=torch.float32, device=B.device)
    dbs = []
    for b, S in zip(biases, Ss):
        dbs.append(None if b is None else (torch.empty_like(b) if S is not None else torch.zeros_like(b)))
    dv = _check_dev(B, dimp_all, *[t for imp in imps for t in imp], *[t for t in list(dAs) + list(Ss) if t is not None])
    itab = _ptr_table([imp[j] if j < J else None for imp in imps for j in range(3)])
    dtab = _ptr_table([dimp_all[i, j] if j < J else None for i in range(nb) for j in range(3)])
    _call('istgcn_fold_bwd_batch', nb, _ptr(B), J, itab, _ptr_table(biases), _ptr_table(dAs), _ptr_table(Ss), dtab,
          _ptr_table([d if S is not None else None for d, S in zip(dbs, Ss)]), _int_array(Cs), K, V, _stream(B), dev=dv)
    return [[dimp_all[i, j] for j in range(J)] for i in range(nb)], dbs


def tcn_fold_fwd(w1, w2, w3, b1, b2, b3, mst, scale):
    """-> (taps [15,Co,Ci], bias [Co]): the 3/9/15-tap branches pre-summed into one 15-tap convolution."""
    Co, Ci = w3.shape[0], w3.shape[1]
    taps = torch.empty((15, Co, Ci), dtype=torch.float32, device=w3.device)
    bias = torch.empty((Co,), dtype=torch.float32, device=w3.device)
    dv = _check_dev(w1, w2, w3, b1, b2, b3, mst, taps, bias)
    _call('istgcn_tcn_fold_fwd', _ptr(w1), _ptr(w2), _ptr(w3), _ptr(b1), _ptr(b2), _ptr(b3), _ptr(mst),
          ctypes.c_float(scale), _ptr(taps), _ptr(bias), Co, Ci, _stream(w3), dev=dv)
    return taps, bias


def tcn_fold_bwd(dtaps, dbias, w1, w2, w3, b1, b2, b3, mst, scale):
    """-> (dw1, dw2, dw3, db1, db2, db3, dmst) in the parameters' own layouts."""
    Co, Ci = w3.shape[0], w3.shape[1]
    outs = [torch.empty_like(t) for t in (w1, w2, w3, b1, b2, b3)]
    dmst = torch.zeros_like(mst)
    dv = _check_dev(dtaps, dbias, w1, w2, w3, b1, b2, b3, mst, dmst, *outs)
    _call('istgcn_tcn_fold_bwd', _ptr(dtaps), _ptr(dbias), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(b1), _ptr(b2), _ptr(b3),
          _ptr(mst), ctypes.c_float(scale), *[_ptr(t) for t in outs], _ptr(dmst), Co, Ci, _stream(w3), dev=dv)
    return (*outs, dmst)


# ----------------------------------------------------------------------------------------------
# optimizer (optim.hip)
# ----------------------------------------------------------------------------------------------
def sgd_step(params, grads, momentum_buf, lr, momentum, weight_decay, nesterov, grad_scale=1.0, found_inf=None, skip_if=None):
    """istgcn_sgd_step over three flat fp32 buffers of equal length (in place on params / momentum_buf).  found_inf:
    int32[1] device tensor raised by a non-finite gradient (those elements are not applied) or None.  skip_if: int32[1]
    device tensor; non-zero at launch = the whole step is skipped (the flag `grad_nonfinite` computed) or None."""
    n = params.numel()
    assert grads.numel() == n and momentum_buf.numel() == n
    assert params.dtype == grads.dtype == momentum_buf.dtype == torch.float32
    for f in (found_inf, skip_if):
        assert f is None or (f.dtype == torch.int32 and f.numel() == 1)
    dv = _check_dev(params, grads, momentum_buf, found_inf, skip_if)
    _call('istgcn_sgd_step', _ptr(params), _ptr(grads), _ptr(momentum_buf), ctypes.c_longlong(n), ctypes.c_float(lr),
          ctypes.c_float(momentum), ctypes.c_float(weight_decay), int(bool(nesterov)), ctypes.c_float(grad_scale),
          _ptr(found_inf), _ptr(skip_if), _stream(params), work=(5.0 * n, 20.0 * n), dev=dv)
    bump_weights_epoch()                  # parameters written through raw pointers: cached inference plans are stale


def grad_nonfinite(grads, flag):
    """istgcn_grad_nonfinite: flag (int32[1], zeroed by the caller) |= any(!isfinite(grads)); no host sync."""
    assert grads.dtype == torch.float32 and flag.dtype == torch.int32 and flag.numel() == 1
    dv = _check_dev(grads, flag)
    _call('istgcn_grad_nonfinite', _ptr(grads), ctypes.c_longlong(grads.numel()), _ptr(flag), _stream(grads),
          work=(float(grads.numel()), 4.0 * grads.numel()), dev=dv)


# ----------------------------------------------------------------------------------------------
# input stage (input.hip): feeder augmentation + data_bn + layout change
# ----------------------------------------------------------------------------------------------
def _aug_args(raw, shift, move, T):
    N, C, Traw, V, M = raw.shape
    if T is None:
        T = Traw
    assert raw.dtype == torch.float32
    if shift is not None:
        assert shift.shape == (N,) and shift.dtype == torch.int32
    if move is not None:
        assert move.shape == (N, T, 6) and move.dtype == torch.float64
    return N, C, Traw, V, M, T


def feeder_augment(raw, shift=None, move=None, T=None):
    """istgcn_feeder_augment -> (N, C, T, V, M) fp32 clips as feeder/tools.py would have produced them."""
    N, C, Traw, V, M, T = _aug_args(raw, shift, move, T)
    out = torch.empty((N, C, T, V, M), dtype=torch.float32, device=raw.device)
    dv = _check_dev(raw, shift, move, out)
    _call('istgcn_feeder_augment', _ptr(raw), _ptr(shift), _ptr(move), _ptr(out), N, C, Traw, T, V, M, _stream(raw), dev=dv)
    return out


def input_stats(raw, stats, shift=None, move=None, T=None):
    N, C, Traw, V, M, T = _aug_args(raw, shift, move, T)
    assert stats.dtype == torch.float64 and stats.shape[-2:] == (2, V * C)
    dv = _check_dev(raw, shift, move, stats)
    _call('istgcn_input_stats', _ptr(raw), _ptr(shift), _ptr(move), _ptr(stats), stats.shape[0], N, C, Traw, T, V, M,
          _stream(raw), dev=dv)


def input_apply(raw, coef, dtype, shift=None, move=None, T=None):
    """-> [N*M, T, V, C] activation (dtype): BatchNorm1d affine coef [>=2][V*C] + the permutes of st_gcnold.py:75-80."""
    N, C, Traw, V, M, T = _aug_args(raw, shift, move, T)
    assert coef.dtype == torch.float32 and coef.shape[1] == V * C and coef.shape[0] >= 2
    out = torch.empty((N * M, T, V, C), dtype=dtype, device=raw.device)
    dv = _check_dev(raw, shift, move, coef, out)
    _call('istgcn_input_apply', _ptr(raw), _ptr(shift), _ptr(move), _ptr(coef), _ptr(out), N, C, Traw, T, V, M,
          _DT[dtype], _stream(raw), dev=dv)
    return out


def input_bwd(raw, dout, coef, stats, shift=None, move=None, T=None):
    N, C, Traw, V, M, T = _aug_args(raw, shift, move, T)
    assert dout.shape == (N * M, T, V, C) and coef.shape == (4, V * C)
    dv = _check_dev(raw, shift, move, dout, coef, stats)
    _call('istgcn_input_bwd', _ptr(raw), _ptr(shift), _ptr(move), _ptr(dout), _ptr(coef), _ptr(stats), stats.shape[0],
          N, C, Traw, T, V, M, dtype_code(dout), _stream(raw), dev=dv)


# ----------------------------------------------------------------------------------------------
# all weight packs of a model in one launch (pack.hip: istgcn_pack_batch)
# ----------------------------------------------------------------------------------------------
class PackPlan:
    """A table of weight-pack jobs (graph-conv weights, temporal-conv taps forward and per data-gradient phase,
    graph-conv backward weights) that one launch executes: `add_*` registers a job and returns its persistent
    destination tensor, `run()` launches (building / uploading the job table on first use).  The job table holds raw
    pointers into the parameters: the OWNER decides when a plan is stale (net/_model.py keys its plan on every parameter's
    data pointer and strides and builds a new one when a parameter moved -- optimizer re-pointing `.data`, `.to()`)."""

    def __init__(self, dtype, device):
        self.dtype, self.device = dtype, device
        self.jobs = []                 # (kind, src view, dst, extra)
        self._table = None
        self._rebind = None            # position in `jobs` while the sources are being re-pointed (begin_rebind)

    def begin_rebind(self):
        """The parameters moved (an nn.DataParallel replica is a fresh set of tensors on every forward; an optimizer
        re-pointed `.data`): replay the SAME sequence of add_* calls -- each replaces the source of the job registered at
        that position and returns its existing destination (no allocation, geometry unchanged) -- then end_rebind()."""
        self._rebind = 0

    def end_rebind(self):
        if self._rebind != len(self.jobs):
            raise RuntimeError('PackPlan.end_rebind: %d of %d jobs re-pointed' % (self._rebind, len(self.jobs)))
        self._rebind = None
        self._table = None             # the job table holds raw source pointers: rebuilt (and uploaded) by the next run()

    def _add(self, kind, src, extra, nelems):
        if self._rebind is not None:
            k0, s0, dst, e0 = self.jobs[self._rebind]
            if k0 != kind or tuple(s0.shape) != tuple(src.shape) or e0 != extra:
                raise RuntimeError('PackPlan rebind: job %d differs from the one registered first' % self._rebind)
            self.jobs[self._rebind] = (kind, src, dst, extra)
            self._rebind += 1
            return dst
        n = nelems()
        if n < 0:
            raise RuntimeError('istgcn pack geometry: invalid')
        dst = torch.empty(int(n), dtype=self.dtype, device=self.device)
        self.jobs.append((kind, src, dst, extra))
        return dst

    def add_gcn(self, wr):
        """wr [Cout][K][Cin] fp32 view -> dst as ops.pack_gcn_weight"""
        _src_ok(wr)
        cout, K, cin = wr.shape
        return self._add(0, wr, None, lambda: _lib.load().istgcn_pack_gcn_elems(cin, cout, K, _DT[self.dtype]))

    def add_tconv(self, wf, V, tap_off, in_mul, tap_sel=None):
        """wf [taps][Cout][Cin] fp32 view -> dst as ops.pack_tconv_weight"""
        _src_ok(wf)
        if tap_sel is None:
            tap_sel = list(range(wf.shape[0]))
        _, cout, cin = wf.shape
        return self._add(1, wf, (V, list(tap_off), in_mul, list(tap_sel)),
                         lambda: _lib.load().istgcn_pack_tconv_elems(V, cin, cout, len(tap_off), _int_array(tap_off), in_mul, _DT[self.dtype]))

    def add_gcn_wb(self, w3):
        """w3 [K][Cout][Cin] fp32 view -> dst as ops.pack_gcn_wb"""
        _src_ok(w3)
        K, cout, cin = w3.shape
        return self._add(2, w3, None, lambda: _lib.load().istgcn_pack_gcn_bwd_elems(cin, cout, K, _DT[self.dtype]))

    def _build(self):
        lib = _lib.load()
        rb = lib.istgcn_pack_job_bytes()
        buf = ctypes.create_string_buffer(rb * len(self.jobs))
        base = ctypes.addressof(buf)
        starts, total, dc = [], 0, _DT[self.dtype]
        ll = ctypes.c_longlong
        for j, (kind, src, dst, extra) in enumerate(self.jobs):
            rec = ctypes.c_void_p(base + j * rb)
            st = src.stride()
            if kind == 0:
                cout, K, cin = src.shape
                nb = lib.istgcn_pack_job_gcn(rec, _ptr(src), ll(st[0]), ll(st[1]), ll(st[2]), _ptr(dst), cin, cout, K, dc)
            elif kind == 1:
                V, tap_off, in_mul, tap_sel = extra
                _, cout, cin = src.shape
                nb = lib.istgcn_pack_job_tconv(rec, _ptr(src), ll(st[0]), ll(st[1]), ll(st[2]), _int_array(tap_sel), _ptr(dst),
                                               V, cin, cout, len(tap_off), _int_array(tap_off), in_mul, dc)
            else:
                K, cout, cin = src.shape
                nb = lib.istgcn_pack_job_gcn_bwd(rec, _ptr(src), ll(st[0]), ll(st[1]), ll(st[2]), _ptr(dst), cin, cout, K, dc)
            if nb < 0:
                raise RuntimeError('istgcn_pack_job: invalid job %d' % j)
            starts.append(total)
            total += nb
        self._ptrs = [src.data_ptr() for _, src, _, _ in self.jobs]
        self._table = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(self.device)
        self._starts = torch.tensor(starts, dtype=torch.int32).to(self.device)
        self._total = total

    def run(self):
        if not self.jobs:
            return
        if self._table is None:
            self._build()
        _call('istgcn_pack_batch', _ptr(self._table), _ptr(self._starts), len(self.jobs), self._total, _DT[self.dtype],
              _stream(self._table), dev=self.device)
