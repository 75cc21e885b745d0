"""Autograd wiring of the HIP kernels: one `torch.autograd.Function` per fused stage.

* `GraphConvFn`  -- the GCN unit alone (net/utils/tgcn.py:76-89 and variants), used by the unit-level
                   drop-in modules in `net/utils/`.
* `STGCNBlockFn` -- a whole st_gcn block (net/st_gcnold.py:197-203 and variants): GCN unit -> BN -> ReLU ->
                   temporal conv (1 branch, 3 pre-summed branches, or the sqrt(C) bottleneck) -> BN -> Dropout
                   -> + residual -> ReLU, forward and hand-written backward, all through `libistgcn_hip.so`.

Everything *tiny* that depends on learnable parameters (folding importances into one adjacency, pushing the
Conv2d bias through the einsum, pre-summing the Inception-TCN taps) is expressed with ordinary differentiable
torch ops on (K,V,V)/(C,C,k)-sized tensors in `fold_*` below, so the kernels only ever see folded operands and
return gradients w.r.t. those; autograd maps them back to the reference's parameters.
Activations are NTVC ([N*M, T, V, C], see csrc/common.hpp); parameters and all statistics are fp32/fp64.
"""
import os

import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------------------------------
# differentiable host-side folds (tiny tensors)
# --------------------------------------------------------------------------------------------------
def fold_adjacency(kind, A, imps, A2=None, A3=None):
    """One effective adjacency for every GCN-unit variant (the einsum is linear in A):
    plain  A*imp                                   st_gcnold.py:86
    incep  A*imp + A2*imp2 + A3*imp3               st_gcn_msgcn.py:116-117 + inceptionv2_gcn.py:69-80
    3a     A*imp + A**2*imp2 + A**3*imp3           tgcn_multi3_fix_3A.py:86-89 (elementwise powers)"""
    if kind == 'plain':
        return A * imps[0]
    if kind == 'incep':
        return A * imps[0] + A2 * imps[1] + A3 * imps[2]
    if kind == '3a':
        return A * imps[0] + (A * A) * imps[1] + (A * A * A) * imps[2]
    raise KeyError(kind)


def fold_bias_term(bias, A_eff, cout):
    """Conv2d bias pushed through the einsum: bterm[w][c] = sum_k b[k*C+c] * sum_v A_eff[k][v][w]."""
    K = A_eff.shape[0]
    return torch.einsum('kc,kw->wc', bias.view(K, cout), A_eff.sum(1)).contiguous()


class FoldFn(torch.autograd.Function):
    """(A_eff, bterm) = fold(B, bias, *imps) in one launch (and one for the gradients): the fused form of
    fold_adjacency + fold_bias_term above for learnable importances on the GPU."""

    @staticmethod
    def forward(ctx, B, bias, C, *imps):
        imps = [i.contiguous() for i in imps]
        A_eff, bterm = ops.fold_fwd(B, imps, bias, C)
        ctx.save_for_backward(B, bias, *imps)
        ctx.C = C
        return A_eff, bterm

    @staticmethod
    def backward(ctx, dA, dbterm):
        B, bias, *imps = ctx.saved_tensors
        dA = dA.contiguous() if dA is not None else None
        S = dbterm.contiguous() if dbterm is not None else None
        dimps, dbias = ops.fold_bwd(B, imps, bias, dA, S, ctx.C)
        return (None, dbias, None) + tuple(dimps)


class FoldAllFn(torch.autograd.Function):
    """FoldFn for ALL blocks of a model at once: (A_eff_0, bterm_0, A_eff_1, bterm_1, ...) = folds(B; biases, importances)
    in one launch, and -- since autograd runs a node's backward only when every output's gradient is known -- all the
    importance / bias gradients in one launch at the very end of the backward pass (were 2 x 10 launches per step).
    forward(ctx, B, Cs, J, *tensors) with tensors = nb biases followed by nb*J importances (block-major)."""

    @staticmethod
    def forward(ctx, B, Cs, J, *tensors):
        nb = len(Cs)
        biases = [None if b is None else b.contiguous() for b in tensors[:nb]]
        imps = [[tensors[nb + i * J + j].contiguous() for j in range(J)] for i in range(nb)]
        A_effs, bts = ops.fold_fwd_batch(B, imps, biases, Cs)
        ctx.save_for_backward(B, *[b for b in biases if b is not None], *[t for imp in imps for t in imp])
        ctx.has_bias = [b is not None for b in biases]
        ctx.Cs, ctx.J = tuple(Cs), J
        out = []
        for a, bt in zip(A_effs, bts):
            out += [a, bt]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        B = saved[0]
        nb, J = len(ctx.Cs), ctx.J
        nbias = sum(ctx.has_bias)
        bl = list(saved[1:1 + nbias])
        biases = [bl.pop(0) if h else None for h in ctx.has_bias]
        flat = saved[1 + nbias:]
        imps = [[flat[i * J + j] for j in range(J)] for i in range(nb)]
        dAs = [None if grads[2 * i] is None else grads[2 * i].contiguous() for i in range(nb)]
        Ss = [None if (grads[2 * i + 1] is None or biases[i] is None) else grads[2 * i + 1].contiguous() for i in range(nb)]
        dimps, dbs = ops.fold_bwd_batch(B, imps, biases, dAs, Ss, ctx.Cs)
        return (None, None, None) + tuple(dbs) + tuple(t for d in dimps for t in d)


def fold_tcn_taps(w1, w2, w3, b1, b2, b3, mst, scale=1.0):
    """x1*m0 + x2*m1 + x3*m2 (st_gcn_multi3_fix_3A_mstcn.py:212-215; /3 in st_gcn_mstcn.py:245) with kernel sizes
    3/9/15 and paddings 1/4/7 is ONE 15-tap convolution: taps [15][Cout][Cin] and one bias."""
    t3 = w3[:, :, :, 0].permute(2, 0, 1) * mst[2]
    t2 = F.pad(w2[:, :, :, 0].permute(2, 0, 1) * mst[1], (0, 0, 0, 0, 3, 3))
    t1 = F.pad(w1[:, :, :, 0].permute(2, 0, 1) * mst[0], (0, 0, 0, 0, 6, 6))
    taps = (t1 + t2 + t3) * scale
    bias = (b1 * mst[0] + b2 * mst[1] + b3 * mst[2]) * scale
    return taps.contiguous(), bias.contiguous()


class TcnTapsFn(torch.autograd.Function):
    """fold_tcn_taps as one launch each way (forward: taps + bias; backward: all six parameter gradients in their own
    layouts + the branch-importance gradient) for GPU parameters."""

    @staticmethod
    def forward(ctx, w1, w2, w3, b1, b2, b3, mst, scale):
        args = [t.contiguous() for t in (w1, w2, w3, b1, b2, b3, mst)]
        ctx.save_for_backward(*args)
        ctx.scale = float(scale)
        return ops.tcn_fold_fwd(*args, ctx.scale)

    @staticmethod
    def backward(ctx, dtaps, dbias):
        w1, w2, w3, b1, b2, b3, mst = ctx.saved_tensors
        if dtaps is None:
            dtaps = torch.zeros((15,) + tuple(w3.shape[:2]), dtype=torch.float32, device=w3.device)
        if dbias is None:
            dbias = torch.zeros_like(b3)
        g = ops.tcn_fold_bwd(dtaps.contiguous(), dbias.contiguous(), w1, w2, w3, b1, b2, b3, mst, ctx.scale)
        return g + (None,)


def fold_tcn_taps_any(w1, w2, w3, b1, b2, b3, mst, scale=1.0):
    """The fused fold when everything is an fp32 GPU tensor with biases, else the torch-op specification."""
    ts = (w1, w2, w3, b1, b2, b3, mst)
    if all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in ts) and \
            w1.shape[2:] == (3, 1) and w2.shape[2:] == (9, 1) and w3.shape[2:] == (15, 1):
        return TcnTapsFn.apply(w1, w2, w3, b1, b2, b3, mst, scale)
    return fold_tcn_taps(w1, w2, w3, b1, b2, b3, mst, scale)


def _pad_width(w, dt):
    """bottleneck width rounded up to whole 16-byte channel vectors of the storage type"""
    epl = 4 if dt == torch.float32 else 8
    return (w + epl - 1) // epl * epl


def _pad_bneck(Ws, bs, Wt, bt, We, w, wp):
    """zero-padded views of the bottleneck's weights for a width of wp >= w channels: Ws [w][C] -> [wp][C], bs [w] -> [wp],
    Wt [k][w][w] -> [k][wp][wp], bt likewise, We [C][w] -> [C][wp] (biases may be None)"""
    if wp == w:
        return Ws, bs, Wt, bt, We
    p = wp - w
    return (F.pad(Ws, (0, 0, 0, p)), None if bs is None else F.pad(bs, (0, p)), F.pad(Wt, (0, p, 0, p)),
            None if bt is None else F.pad(bt, (0, p)), F.pad(We, (0, p)))


def _bneck_rc(cfg, V, dt, rows=None):
    """Do the register-chained bottleneck kernels (csrc/bneck_rc.hip) serve this block?  16-bit storage, stride <= taps <= 15
    (every stride phase of the data gradient has a tap), stride 1 or 2, V <= 32, 64 / 128 / 256 wide, the wide tensor behind
    one 4 GiB buffer descriptor (`rows` = N*M*T*V of the block's input); `ops.BNECK_RC = False` (ISTGCN_BNECK_RC=0) keeps the
    generic temporal-conv kernels, which is also where every other shape goes."""
    if rows is not None and rows * max(cfg.cout, cfg.cin) * 2 >= (1 << 32):
        return False
    return (ops.BNECK_RC and cfg.stride <= cfg.ksize <= 15 and cfg.stride in (1, 2) and
            ops.bneck_ok(V, cfg.cout, cfg.width, _pad_width(cfg.width, dt), dt))


_EYE = {}
# ISTGCN_WGRAD_SIDE=1: the temporal conv's weight gradient on a side stream next to the data gradient (off by default: 0.1-0.7 %
# of a step on one GPU, and a data-gradient launch that shares the CUs with it no longer has a duration of its own -- the
# per-kernel roofline of bench.py reads 0.27 instead of 0.38 for the same work)
WGRAD_SIDE_STREAM = os.environ.get('ISTGCN_WGRAD_SIDE', '0') == '1'
_SIDE = {}


def _side_stream(device):
    """One side stream per device for launches that overlap the main sequence (see STGCNBlockFn.backward)."""
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st



def _eye(V, device):
    """[1,V,V] identity adjacency (K = 1, A = I turns the graph-conv kernel into a strided 1x1 conv), cached per device."""
    e = _EYE.get((V, device))
    if e is None:
        e = _EYE[(V, device)] = torch.eye(V, device=device, dtype=torch.float32).view(1, V, V)
    return e


def conv_bwd_phases(k, stride):
    """[(phase, tap offsets, tap selection)] of the data gradient's launches (phases without a tap are left out)."""
    out = []
    for phase in range(stride):
        tl = ops.conv_taps_bwd(k, stride, phase)
        if tl:
            out.append((phase, [dj for _, dj in tl], [j for j, _ in tl]))
    return out


def _conv_bwd_data(dz, w_taps, k, stride, T_in, cin, V, aux=None, maux=None, stats=None, packed=None, before_last=None):
    """Data gradient of a (k,1)/stride conv with taps w_taps [k][Cout][Cin]: one tconv launch per output phase.
    packed: {phase: fragment-packed transposed taps} from a PackPlan (else packed here, one launch per phase).
    before_last: called right before the LAST launch that adds into `stats` (arms the BatchNorm tail, ops.bn_bwd_coef(defer=True))."""
    NM, Tz = dz.shape[0], dz.shape[1]
    out = torch.empty((NM, T_in, V, cin), dtype=dz.dtype, device=dz.device)
    launching = [ph for ph in range(stride) if (T_in - ph + stride - 1) // stride > 0 and ops.conv_taps_bwd(k, stride, ph)]
    for phase in range(stride):
        tl = ops.conv_taps_bwd(k, stride, phase)
        Mlog = (T_in - phase + stride - 1) // stride
        if Mlog <= 0:
            continue
        if not tl:                       # no tap lands on this phase (k < stride): those frames get no gradient
            out[:, phase::stride].zero_()
            continue
        if before_last is not None and phase == launching[-1]:
            before_last()
        offs = [dj for _, dj in tl]
        # transposed taps [k][Cin][Cout] as a view; the packer gathers this phase's taps straight from the parameter
        wp = packed[phase] if packed is not None else \
            ops.pack_tconv_weight(w_taps.transpose(1, 2), V, offs, 1, dz.dtype, tap_sel=[j for j, _ in tl])
        ops.tconv(dz, wp, cin, offs, aux=aux, maux=maux, out=out, stats=stats, mode=0 if aux is None else 1,
                  Tout=T_in, Mlog=Mlog, in_mul=1, out_mul=stride, out_off=phase)
    return out


# --------------------------------------------------------------------------------------------------
# GCN unit
# --------------------------------------------------------------------------------------------------
class GraphConvFn(torch.autograd.Function):
    """y = einsum('nkctv,kvw->nctw', conv1x1(x; W, b), A_eff) on NTVC tensors.
    inputs: x [NM,T,V,Cin], A_eff [K,V,V], bterm [V,Cout] (or None), W3 [K,Cout,Cin]; nnz_cap int; pattern [K,V,V]
    fp32 or None = dense (every entry of A_eff gets its gradient, as autograd of net/utils/tgcn.py:86 gives)."""

    @staticmethod
    def forward(ctx, x, A_eff, bterm, W3, nnz_cap, pattern=None):
        K, cout, cin = W3.shape
        A_eff = A_eff.contiguous()
        wp = ops.pack_gcn_weight(W3.permute(1, 0, 2), x.dtype)
        y = ops.gcn_forward(x, A_eff, wp, cout, bterm=bterm, nnz_cap=nnz_cap)
        ctx.save_for_backward(x, A_eff, W3)
        ctx.nnz_cap = nnz_cap
        ctx.pattern = pattern
        ctx.has_b = bterm is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, A_eff, W3 = ctx.saved_tensors
        K, cout, cin = W3.shape
        dy = dy.contiguous()
        need_A = ctx.needs_input_grad[1]
        dW, S = ops.gcn_wgrad(dy, x, A_eff, want_S=ctx.has_b, nnz_cap=ctx.nnz_cap)
        dx = dA = None
        if ctx.needs_input_grad[0] or need_A:
            pat, cap = ctx.pattern, ctx.nnz_cap
            if need_A and pat is None:
                pat, cap = torch.ones_like(A_eff), A_eff.numel()
            dx, dA = ops.gcn_bwd_data(dy, A_eff, W3, x=x, want_dA=need_A, nnz_cap=cap, pattern=pat)
        return dx, dA, (S if ctx.has_b else None), dW, None, None


# --------------------------------------------------------------------------------------------------
# whole st_gcn block
# --------------------------------------------------------------------------------------------------
class BlockCfg:
    """Static description of one block (shapes, variant, BatchNorm hyper-parameters)."""

    def __init__(self, cin, cout, K, V, stride, residual, tcn, ksize, p_drop, nnz_cap, width=None,
                 momentum=0.1, eps=1e-5, pattern=None):
        self.cin, self.cout, self.K, self.V, self.stride = cin, cout, K, V, stride
        self.residual = residual          # 'none' | 'id' | 'conv'
        self.tcn = tcn                    # 'conv' (single or pre-summed multi-branch) | 'bneck'
        self.ksize = ksize                # temporal taps of the (folded) conv: 9 or 15
        self.p_drop, self.nnz_cap = p_drop, nnz_cap
        self.width = width                # bottleneck width int(sqrt(C))
        self.momentum, self.eps = momentum, eps
        self.pattern = pattern            # [K,V,V] fp32 sparsity pattern of the adjacency gradient (None: dense)
        self.packed = None                # dict of fragment-packed weights from the Model's PackPlan (None: pack per call)
        self.seed_epoch = None            # int64[1] device tensor added to the dropout seed at kernel run time (graph replay)
        self.step_arena = None            # ops.StepArena of the current forward (the Model's one zero fill per step) or None


class STGCNBlockFn(torch.autograd.Function):
    """forward(cfg, training, seed, bn_buffers, x, A_eff, bterm, Wg3, g1, b1, Wt, bt, g2, b2,
               Wr, br, gr, betar, Ws, bs, We, be)
    Wt/bt: temporal taps [k][Cw][Cw] + bias (Cw = cout, or the bottleneck width); Ws/bs, We/be: the bottleneck's
    1x1 in/out projections (None otherwise); Wr/br/gr/betar: the strided 1x1 residual conv + its BatchNorm.
    bn_buffers: dict name -> (running_mean, running_var) updated in place when training."""

    @staticmethod
    def forward(ctx, cfg, training, seed, bufs, x, A_eff, bterm, Wg3, g1, b1, Wt, bt, g2, b2,
                Wr, br, gr, betar, Ws, bs, We, be):
        dt = x.dtype
        NM, T, V, cin = x.shape
        cout, s = cfg.cout, cfg.stride
        Tz = (T - 1) // s + 1
        dev = x.device
        A_eff = A_eff.contiguous()
        # 1. graph conv (+ BN1 batch sums)
        st1 = ops.stats_scratch(0, cout, dev) if training else None
        pk = cfg.packed or {}
        wp = pk['wg'] if 'wg' in pk else ops.pack_gcn_weight(Wg3.permute(1, 0, 2), dt)
        # (training: the BatchNorm arithmetic is armed as the tail of the kernel that produces the sums -- its last workgroup
        #  finalises -- and bn_tail_flush() launches the stand-alone kernel only if that kernel's variant has no tail)
        coef1 = ops.bn_finalize(st1, NM * T * V, g1, b1, bufs['bn1'][0], bufs['bn1'][1], cfg.momentum, cfg.eps, training,
                                clear=True, defer=True) if training else None
        g = ops.gcn_forward(x, A_eff, wp, cout, bterm=bterm, stats=st1, nnz_cap=cfg.nnz_cap)
        if training:
            ops.bn_tail_flush()
        else:
            coef1 = ops.bn_finalize(st1, NM * T * V, g1, b1, bufs['bn1'][0], bufs['bn1'][1], cfg.momentum, cfg.eps, training,
                                    clear=True)
        # 2. temporal conv (BN1+ReLU fused into the staging; BN2 batch sums from the epilogue)
        st2 = ops.stats_scratch(1, cout, dev) if training else None
        coef2 = ops.bn_finalize(st2, NM * Tz * V, g2, b2, bufs['bn2'][0], bufs['bn2'][1], cfg.momentum, cfg.eps, training,
                                clear=True, defer=True) if training else None
        taps, in_mul = ops.conv_taps_fwd(cfg.ksize, s)
        q = yb = None
        if cfg.tcn == 'conv':
            wt = pk['wt'] if 'wt' in pk else ops.pack_tconv_weight(Wt, V, taps, in_mul, dt)
            z = ops.tconv(g, wt, cout, taps, bias=bt, pre=coef1[:2].contiguous(), pre_relu=True, stats=st2,
                          Tout=Tz, Mlog=Tz, in_mul=in_mul)
        else:
            # the bottleneck width int(sqrt(C)) (st_gcn_mstcn_1x1.py:190-224) is 11 at 128 channels: 22-byte rows that no
            # 16-byte vector path can touch (measured, config 5: 4.3 ms per 128-channel block against 1.8 / 1.2 ms for the
            # 8- and 16-wide ones).  The narrow tensors are therefore stored with the width padded to whole vectors; the
            # padding channels carry zero weights and biases, so they ARE zeros and change nothing downstream.
            w, wp = cfg.width, _pad_width(cfg.width, dt)
            if _bneck_rc(cfg, V, dt, NM * T * V):
                # 16-bit storage: two register-chained stream kernels (csrc/bneck_rc.hip) -- wide -> narrow, then
                # narrow -> 15 taps -> narrow (saved) -> wide with the BatchNorm sums; weights read in place, no packs
                q = ops.bneck_in(g, Ws, wp, bias=bs, pre=coef1[:2].contiguous(), pre_relu=True)
                yb, z = ops.bneck_out(q, Wt, list(range(len(taps))), taps[0], We, cout, bt=bt, be=be, stats=st2, mode=0,
                                      Tout=Tz, Mlog=Tz, in_mul=in_mul)
            else:
                Ws_, bs_, Wt_, bt_, We_ = _pad_bneck(Ws, bs, Wt, bt, We, w, wp)
                ws = ops.pack_tconv_weight(Ws_.view(1, wp, cout), V, [0], 1, dt)
                q = ops.tconv(g, ws, wp, [0], bias=bs_, pre=coef1[:2].contiguous(), pre_relu=True, Tout=T, Mlog=T)
                wt = ops.pack_tconv_weight(Wt_, V, taps, in_mul, dt)
                yb = ops.tconv(q, wt, wp, taps, bias=bt_, Tout=Tz, Mlog=Tz, in_mul=in_mul)
                we = ops.pack_tconv_weight(We_.view(1, cout, wp), V, [0], 1, dt)
                z = ops.tconv(yb, we, cout, [0], bias=be, stats=st2, Tout=Tz, Mlog=Tz)
        if training:
            ops.bn_tail_flush()
        else:
            coef2 = ops.bn_finalize(st2, NM * Tz * V, g2, b2, bufs['bn2'][0], bufs['bn2'][1], cfg.momentum, cfg.eps, training,
                                    clear=True)
        # 3. residual branch
        r = coefr = None
        if cfg.residual == 'id':
            res, cr = x, None
        elif cfg.residual == 'conv':
            strs = ops.stats_scratch(2, cout, dev) if training else None
            if ops.gcn_rc_serves(cin, cout, 1, V, dt):
                # 16-bit storage: the strided 1 x 1 conv as the register-chained graph-conv kernel with K = 1, A = I (the form its
                # data gradient has had since round 3): 57 us against 83 on `tconv`, whose item structure is built for 9 taps
                wrg = pk['wrg'] if 'wrg' in pk else ops.pack_gcn_weight(Wr.unsqueeze(1), dt)
                bt_r = None if br is None else br.view(1, cout).expand(V, cout).contiguous()
                r = ops.gcn_forward(x, _eye(V, dev), wrg, cout, bterm=bt_r, stats=strs, Tout=Tz, in_t_stride=s, nnz_cap=V)
            else:
                wr = pk['wr'] if 'wr' in pk else ops.pack_tconv_weight(Wr.view(1, cout, cin), V, [0], s, dt)
                r = ops.tconv(x, wr, cout, [0], bias=br, stats=strs, Tout=Tz, Mlog=Tz, in_mul=s)
            coefr = ops.bn_finalize(strs, NM * Tz * V, gr, betar, bufs['bnr'][0], bufs['bnr'][1], cfg.momentum,
                                    cfg.eps, training, clear=True)
            res, cr = r, coefr[:2].contiguous()
        else:
            res, cr = None, None
        # 4. BN2 + dropout + residual + ReLU
        p = cfg.p_drop if training else 0.0
        # (training: the ReLU mask of `out` as one byte per vector, so the backward does not read `out` again)
        out, rmask = ops.block_out_fwd(z, coef2[:2].contiguous(), res, cr, p, seed, epoch=cfg.seed_epoch, want_mask=True)
        ctx.cfg, ctx.training, ctx.seed, ctx.p = cfg, training, seed, p
        ctx.step_arena = cfg.step_arena   # captured now: the next forward replaces cfg.step_arena before this backward runs
        ctx.save_for_backward(x, A_eff, Wg3, g1, Wt, g2, Wr, gr, Ws, We, g, z, out, coef1, coef2, r, coefr, q, yb, rmask)
        ctx.has_b = bterm is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg, training, seed, p = ctx.cfg, ctx.training, ctx.seed, ctx.p
        (x, A_eff, Wg3, g1, Wt, g2, Wr, gr, Ws, We, g, z, out, coef1, coef2, r, coefr, q, yb, rmask) = ctx.saved_tensors
        dt = x.dtype
        NM, T, V, cin = x.shape
        cout, s, k = cfg.cout, cfg.stride, cfg.ksize
        Tz = z.shape[1]
        dout = dout.contiguous()
        if dout.dtype != dt:
            dout = dout.to(dt)
        pk = cfg.packed or {}
        # 4'. ReLU + residual split, BatchNorm-backward sums of tcn.3 (and of the residual BN)
        # (bottleneck blocks at 64 / 128 channels in 16-bit storage: dz is never written -- the stream kernel that consumes it
        #  forms it in registers from dres and z, dropout mask included; ops.BNECK_FUSE_BN = False keeps the separate pass)
        fuse_in = (cfg.tcn == 'bneck' and ops.BNECK_FUSE_BN and _bneck_rc(cfg, V, dt, NM * T * V) and
                   ops.bneck_bwd_in_ok(cout, cfg.width, _pad_width(cfg.width, dt), dt))
        # Round 5: dres = dout * [out > 0] is not written where every reader can take dout and the forward's 1-bit-per-element
        # ReLU mask itself -- `affine2` (tcn.3's and the residual BatchNorm's backward) always can, the identity-residual addend
        # of the graph conv's data gradient where the register-chained kernel serves it; the fused bottleneck stream kernel
        # reads dres (one tensor write less per block: ops.DRES_FREE = False / ISTGCN_DRES_FREE=0 keeps the tensor)
        dres_free = (ops.DRES_FREE and rmask is not None and not fuse_in and
                     (cfg.residual != 'id' or ops.gcn_bwd_addend_mask_ok(V, cin, cout, A_eff.shape[0], dt)))
        dres, st2b, strb, (abc2, dg2, db2) = ops.block_out_bwd(dout, out, z, coef2, r, coefr, p, seed, scratch=True,
                                                               epoch=cfg.seed_epoch, relu_mask=rmask,
                                                               tail=(NM * Tz * V, g2, training), want_dres=not dres_free)
        dsrc, dmask = (dout, rmask) if dres_free else (dres, None)       # dres as (tensor, mask) for its readers
        dz = None if fuse_in else ops.affine2(dsrc, z, abc2, p, seed, epoch=cfg.seed_epoch, relu_mask=dmask)
        # 2'. temporal conv: weight gradient + data gradient (ReLU mask of BN1 and its backward sums fused)
        taps, in_mul = ops.conv_taps_fwd(k, s)
        pre1 = coef1[:2].contiguous()
        st1b = ops.stats_scratch(2, cout, x.device)
        bwd1 = []                          # (abc1, dgamma1, dbeta1) of tcn.0's BatchNorm: armed as the tail of the last launch that adds into st1b

        def arm1():
            bwd1.append(ops.bn_bwd_coef(st1b, NM * T * V, g1, coef1, training, clear=True, defer=True))
        dWs = dbs = dWe = dbe = None
        need_A = ctx.needs_input_grad[5]
        K = A_eff.shape[0]
        # all fp32 gradient accumulators of this block out of ONE zero-filled allocation
        cw = cout if cfg.tcn == 'conv' else cfg.width
        shapes = [(len(taps), cw, cw), (cw,), (K, cout, cin)]
        if ctx.has_b:
            shapes.append((V, cout))
        if need_A:
            shapes.append((K, V, V))
        if cfg.residual == 'conv':
            shapes += [(1, cout, cin), (cout,)]
        arena = ops.ZeroArena(shapes, x.device, parent=ctx.step_arena)
        buf_t = (arena.take(), arena.take())
        buf_g = (arena.take(), arena.take() if ctx.has_b else None)
        buf_A = arena.take() if need_A else None
        buf_r = (arena.take(), arena.take()) if cfg.residual == 'conv' else None
        side = None
        if cfg.tcn == 'conv':
            # (training: the conv's bias feeds a batch-statistics BatchNorm, so sum_p dz = 0 identically -- the reference's
            #  autograd returns the rounding noise of that sum, 1e-5 next to weight gradients of 1e+2 in the fixtures; the
            #  column sums are not computed and the gradient is the zero-filled buffer)
            # The weight gradient runs on a SIDE stream next to the data gradient and everything behind it on this one (both
            # read dz, neither reads the other): its last ~30 us -- 37.7 MB of per-CU partial sums leaving the chip, matrix
            # cores idle -- overlap the next kernels instead of standing in their way.  Joined before the gradients are returned.
            # (not under hipGraph capture: the cross-stream edges cost a replayed step 0.14 ms, measured)
            use_side = WGRAD_SIDE_STREAM and not torch.cuda.is_current_stream_capturing()
            side = _side_stream(x.device) if use_side else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):
                    dWt, dbt = ops.tconv_wgrad(dz, g, taps, in_mul=in_mul, pre=pre1, pre_relu=True, out=buf_t, want_bias=not training)
            else:
                dWt, dbt = ops.tconv_wgrad(dz, g, taps, in_mul=in_mul, pre=pre1, pre_relu=True, out=buf_t, want_bias=not training)
            d1 = _conv_bwd_data(dz, Wt, k, s, T, cout, V, aux=g, maux=coef1, stats=st1b, packed=pk.get('wt_bwd'),
                                before_last=arm1)
        else:
            w, wp = cfg.width, _pad_width(cfg.width, dt)
            rc = _bneck_rc(cfg, V, dt, NM * T * V)
            if not rc:
                Ws_, _, Wt_, _, We_ = _pad_bneck(Ws, None, Wt, None, We, w, wp)
            if fuse_in:
                dyb, dWe3, dbe = ops.bneck_bwd_in(dres, z, abc2, yb, We.t(), wp, p_drop=p, seed=seed, epoch=cfg.seed_epoch)
            elif rc:
                dWe3, dbe = ops.bneck_wgrad(dz, yb, True)              # [cout][wp], [cout]
            else:
                dWe3, dbe = ops.tconv_wgrad(dz, yb, [0], in_mul=1)
            dWe = dWe3.view(cout, wp)[:, :w]
            if fuse_in:
                pass
            elif rc:
                dyb = ops.bneck_in(dz, We.t(), wp)                       # dyb = We^T dz (the transposed view is read in place)
            else:
                dyb = _conv_bwd_data(dz, We_.view(1, cout, wp), 1, 1, Tz, wp, V)
            if rc:
                dWt, dbt = ops.bneck_wgrad_taps(dyb, q, len(taps), taps[0], in_mul=in_mul)    # [k][wp][wp], [wp]
            else:
                dWt, dbt = ops.tconv_wgrad(dyb, q, taps, in_mul=in_mul, out=buf_t if wp == w else None)
            if rc:
                # dq = sum_j Wt_j^T dyb and d1 = [relu mask] Ws^T dq with the BatchNorm-backward sums: one launch per stride
                # phase, the taps of a phase in ascending order of the dyb frame they read
                dq = torch.empty((NM, T, V, wp), dtype=dt, device=dyb.device)
                d1 = torch.empty((NM, T, V, cout), dtype=dt, device=dyb.device)
                for phase in range(s):
                    tl = sorted(ops.conv_taps_bwd(k, s, phase), key=lambda jd: jd[1])
                    if phase == s - 1:
                        arm1()
                    ops.bneck_out(dyb, Wt.transpose(1, 2), [j for j, _ in tl], tl[0][1], Ws.t(), cout, aux=g, maux=coef1,
                                  stats=st1b, mode=1, Tout=T, Mlog=(T - phase + s - 1) // s, in_mul=1, out_mul=s,
                                  out_off=phase, yb=dq, z=d1)
            else:
                dq = _conv_bwd_data(dyb, Wt_, k, s, T, wp, V)
            if rc:
                dWs3, dbs = ops.bneck_wgrad(g, dq, False, pre=pre1, pre_relu=True)   # [wp][cout], [wp]
            else:
                dWs3, dbs = ops.tconv_wgrad(dq, g, [0], in_mul=1, pre=pre1, pre_relu=True)
            dWs = dWs3.view(wp, cout)[:w]
            if not rc:
                d1 = _conv_bwd_data(dq, Ws_.view(1, wp, cout), 1, 1, T, cout, V, aux=g, maux=coef1, stats=st1b, before_last=arm1)
            if wp != w:
                dWt, dbt, dbs = dWt[:, :w, :w], dbt[:w], dbs[:w]
            # (the register-chained kernels write their own zero-filled gradient buffers; the arena slots stay unused)
        if not bwd1:                       # no launch added into st1b (an empty time axis): the coefficients of all-zero sums
            arm1()
        ops.bn_tail_flush()
        abc1, dg1, db1 = bwd1[0]
        dg = ops.affine2(d1, g, abc1)
        # 1'. graph conv: parameter gradients, then the data gradient with the residual gradient folded in
        dWg, S = ops.gcn_wgrad(dg, x, A_eff, want_S=ctx.has_b, nnz_cap=cfg.nnz_cap, out=buf_g)
        dWr = dbr = dgr = dbetar = None
        dx = dA = None
        if ctx.needs_input_grad[4] or need_A or cfg.residual == 'conv':
            addend = dsrc if cfg.residual == 'id' else None
            pat, cap = cfg.pattern, cfg.nnz_cap
            if need_A and pat is None:
                pat, cap = torch.ones_like(A_eff), A_eff.numel()
            # (first block: the input needs no gradient -- only the adjacency gradient is computed where the kernel has that form)
            dx, dA = ops.gcn_bwd_data(dg, A_eff, Wg3, x=x, addend=addend, want_dA=need_A, nnz_cap=cap, dA_out=buf_A,
                                      pattern=pat, wb=pk.get('wb'), addend_mask=dmask if addend is not None else None,
                                      want_dx=ctx.needs_input_grad[4] or cfg.residual == 'conv')
        if cfg.residual == 'conv':
            abcr, dgr, dbetar = ops.bn_bwd_coef(strb, NM * Tz * V, gr, coefr, training, clear=True)
            dr = ops.affine2(dsrc, r, abcr, relu_mask=dmask)
            dWr3, dbr = ops.tconv_wgrad(dr, x, [0], in_mul=s, out=buf_r, want_bias=not training)     # (as for tcn.2 above)
            dWr = dWr3.view(cout, cin)
            eye = _eye(V, x.device)
            wrt = pk['wrt'] if 'wrt' in pk else ops.pack_gcn_weight(Wr.t().unsqueeze(1), dt)     # [cin][1][cout] view
            ops.gcn_forward(dr, eye, wrt, cin, addend=dx, out=dx, Tout=T, out_t_stride=s, nnz_cap=V)
        if cfg.tcn == 'conv' and side is not None:
            torch.cuda.current_stream(x.device).wait_stream(side)
        return (None, None, None, None, dx, dA, (S if ctx.has_b else None), dWg, dg1, db1, dWt, dbt, dg2, db2,
                dWr, dbr, dgr, dbetar, dWs, dbs, dWe, dbe)


class PoolFn(torch.autograd.Function):
    """Trunk output [N*M, T, V, C] -> clip features [N, C] fp32: F.avg_pool2d over (T, V), then the mean over the M persons
    (net/st_gcnold.py:89-91).  Two launches forward (partial sums + one reduction), one backward (broadcast store)."""

    @staticmethod
    def forward(ctx, y, M):
        ctx.meta = (tuple(y.shape), y.dtype, int(M))
        return ops.pool_fwd(y.contiguous(), int(M))

    @staticmethod
    def backward(ctx, dfeat):
        shape, dtype, M = ctx.meta
        return ops.pool_bwd(dfeat.float(), shape, dtype, M), None


# --------------------------------------------------------------------------------------------------
# input stage: (feeder augmentation +) data_bn + layout change
# --------------------------------------------------------------------------------------------------
class InputStageFn(torch.autograd.Function):
    """x (N,C,T,V,M) fp32 -> data_bn (BatchNorm1d over the V*C channels v*C+c, net/st_gcnold.py:74-80) -> NTVC activation
    [N*M, T', V, C] in `dtype`, optionally through the feeder's augmentation (feeder/tools.py:31-101: frame shift + zero
    padding, per-frame affine of channels 0,1).  Two launches in training (batch sums, apply), one in eval; the backward
    is one reduction for d(data_bn.weight), d(data_bn.bias) -- the clip itself needs no gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, dtype, shift, move, T):
        x = x.contiguous()
        N, C, Traw, V, M = x.shape
        T = Traw if T is None else T
        st = None
        if training:
            st = ops.stats_scratch(3, V * C, x.device)
            ops.input_stats(x, st, shift, move, T)
        coef = ops.bn_finalize(st, N * M * T, gamma, beta, running_mean, running_var, momentum, eps, training, clear=True)
        y = ops.input_apply(x, coef, dtype, shift, move, T)
        ctx.save_for_backward(x, gamma, coef, shift, move)
        ctx.training, ctx.T = training, T
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, coef, shift, move = ctx.saved_tensors
        N, C, Traw, V, M = x.shape
        dgamma = dbeta = None
        if gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            st = ops.stats_scratch(3, V * C, x.device)
            ops.input_bwd(x, dy.contiguous(), coef, st, shift, move, ctx.T)
            _, dgamma, dbeta = ops.bn_bwd_coef(st, N * M * ctx.T, gamma, coef, ctx.training, clear=True)
        return (None, dgamma, dbeta) + (None,) * 9


# --------------------------------------------------------------------------------------------------
# inference path: eval-mode BatchNorm is affine -> folded into the neighbouring weights, nothing saved
# --------------------------------------------------------------------------------------------------
def _bn_affine(bn):
    """(scale, shift) of an eval-mode BatchNorm from its running statistics: y = x*scale + shift."""
    gamma, beta, rm, rv, eps = bn
    scale = (gamma if gamma is not None else 1.0) * torch.rsqrt(rv + eps)
    shift = (beta if beta is not None else 0.0) - rm * scale
    return scale, shift


def build_infer_plan(cfg, dt, A_eff, bterm, Wg3, bn1, Wt, bt, bn2, Wr, br, bnr, Ws, bs, We, be):
    """Constants of one st_gcn block for eval-mode inference (net/st_gcnold.py:197-203 under model.eval(), as
    processor/recognition.py:347-348 and the demos processor/demo_offline.py:68-98 run it):
      tcn.0 (BatchNorm after the graph conv)   -> the temporal conv's `pre` affine (+ReLU), as in training
      tcn.3 (BatchNorm after the temporal conv) -> folded into the temporal conv's weights and bias
      residual BatchNorm                        -> folded into the strided 1x1 residual conv's weights and bias
      Dropout                                   -> identity
    so the block is graph conv + ONE temporal-conv launch whose epilogue adds the residual and applies the ReLU.
    bnX = (gamma, beta, running_mean, running_var, eps).  Returns a dict of packed weights (rebuilt only when a
    parameter changes: net/_model.py caches it on the parameters' version counters)."""
    V, s, cout, cin = cfg.V, cfg.stride, cfg.cout, cfg.cin
    with torch.no_grad():
        s1, h1 = _bn_affine(bn1)
        s2, h2 = _bn_affine(bn2)
        plan = {'A': A_eff.detach().contiguous(), 'bterm': None if bterm is None else bterm.detach().contiguous(),
                'wg': ops.pack_gcn_weight(Wg3.permute(1, 0, 2), dt), 'pre1': torch.stack([s1, h1]).contiguous()}
        taps, in_mul = ops.conv_taps_fwd(cfg.ksize, s)
        zero = torch.zeros((), dtype=torch.float32, device=s2.device)
        if cfg.tcn == 'conv':
            plan['wt'] = ops.pack_tconv_weight((Wt * s2.view(1, -1, 1)).contiguous(), V, taps, in_mul, dt)
            plan['bt'] = ((bt if bt is not None else zero) * s2 + h2).contiguous()
        else:
            w, wp = cfg.width, _pad_width(cfg.width, dt)          # width padded to whole channel vectors, as in training
            Ws_, bs_, Wt_, bt_, We_ = _pad_bneck(Ws, bs, Wt, bt, (We * s2.view(-1, 1)), w, wp)
            plan['wp'] = wp
            plan['ws'] = ops.pack_tconv_weight(Ws_.contiguous().view(1, wp, cout), V, [0], 1, dt)
            plan['bs'] = bs_
            plan['wt'] = ops.pack_tconv_weight(Wt_.contiguous(), V, taps, in_mul, dt)
            plan['bt'] = bt_
            plan['we'] = ops.pack_tconv_weight(We_.contiguous().view(1, cout, wp), V, [0], 1, dt)
            plan['be'] = ((be if be is not None else zero) * s2 + h2).contiguous()
        if cfg.residual == 'conv':
            sr, hr = _bn_affine(bnr)
            plan['wr'] = ops.pack_tconv_weight((Wr * sr.view(-1, 1)).contiguous().view(1, cout, cin), V, [0], s, dt)
            plan['br'] = ((br if br is not None else zero) * sr + hr).contiguous()
    return plan


def run_infer_plan(cfg, plan, x):
    """x [NM,T,V,cin] -> block output [NM,T/stride,V,cout]; 2 launches (3 with a residual conv, 4 with the bottleneck)."""
    NM, T, V, _ = x.shape
    cout, s = cfg.cout, cfg.stride
    Tz = (T - 1) // s + 1
    with torch.no_grad():
        g = ops.gcn_forward(x, plan['A'], plan['wg'], cout, bterm=plan['bterm'], nnz_cap=cfg.nnz_cap)
        res = None
        if cfg.residual == 'id':
            res = x
        elif cfg.residual == 'conv':
            res = ops.tconv(x, plan['wr'], cout, [0], bias=plan['br'], Tout=Tz, Mlog=Tz, in_mul=s)
        taps, in_mul = ops.conv_taps_fwd(cfg.ksize, s)
        if cfg.tcn == 'conv':
            return ops.tconv(g, plan['wt'], cout, taps, bias=plan['bt'], pre=plan['pre1'], pre_relu=True, aux=res,
                             mode=2, Tout=Tz, Mlog=Tz, in_mul=in_mul)
        w = plan['wp']
        q = ops.tconv(g, plan['ws'], w, [0], bias=plan['bs'], pre=plan['pre1'], pre_relu=True, Tout=T, Mlog=T)
        yb = ops.tconv(q, plan['wt'], w, taps, bias=plan['bt'], Tout=Tz, Mlog=Tz, in_mul=in_mul)
        return ops.tconv(yb, plan['we'], cout, [0], bias=plan['be'], aux=res, mode=2, Tout=Tz, Mlog=Tz)
