"""Table-driven implementation behind the drop-in `net.*` modules.

The reference's 21 model files are copy-paste variants of one structure; here one `STGCNBlock`
and one `STGCNModel` are specialised per variant by `make(kind)`, which returns classes named like the reference's
(`Model`, `st_gcn`) with the reference's constructor signatures, forward signatures, attribute names and therefore
`state_dict()` keys/shapes (incl. the dead parameters `linear.*`, `gcn.branch.bn.*`).  Parameters live in real
`nn.Conv2d` / `nn.BatchNorm2d` sub-modules so `model.apply(weights_init)` (processor/recognition.py:31-44,149),
`load_weights`, `DataParallel` and SGD over `model.parameters()` behave as upstream; `forward` never calls those
sub-modules -- it hands their tensors to the HIP kernels through `functional.STGCNBlockFn`.

There is no CPU path: tensors must be on an MI355X, otherwise RuntimeError.
"""
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn
from .. import ops
from .utils.graph import Graph

PLAN10 = ((64, 1), (64, 1), (64, 1), (64, 1), (128, 2), (128, 1), (128, 1), (256, 2), (256, 1), (256, 1))
PLAN7 = ((64, 1), (64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1))
PLAN13 = ((64, 1), (64, 1), (64, 1), (64, 1), (64, 1), (128, 2), (128, 1), (128, 1), (128, 1), (256, 2), (256, 1),
          (256, 1), (256, 1))

# kind -> gcn unit, tcn unit, block plan, dead Linear(3,C) present (st_gcnold.py:178)
VARIANTS = {
    'st_gcnold': dict(gcn='plain', tcn='single', plan=PLAN10, dead_linear=True),
    'st_gcn': dict(gcn='plain', tcn='single', plan=PLAN10, dead_linear=True),      # name the shipped YAMLs use
    'st_gcn_tanh': dict(gcn='plain', tcn='single', plan=PLAN10, dead_linear=True),
    'st_gcn_msgcn': dict(gcn='incep', tcn='single', plan=PLAN10, dead_linear=False),
    'st_gcn_msgcn_new': dict(gcn='incep', tcn='single', plan=PLAN7, dead_linear=False),
    'st_gcn_deep_msgcn': dict(gcn='incep', tcn='single', plan=PLAN13, dead_linear=False),
    'st_gcn_mstcn': dict(gcn='plain', tcn='multi3', plan=PLAN7, dead_linear=False),
    'st_gcn_mstcn_1x1': dict(gcn='plain', tcn='bneck', plan=PLAN10, dead_linear=False),
    'st_gcn_mstcn_1x1_deep': dict(gcn='plain', tcn='bneck', plan=PLAN13, dead_linear=False),
    'st_gcn_multi3_fix_3A_mstcn': dict(gcn='3a', tcn='multi', plan=PLAN10, dead_linear=False),
}

_DTYPES = {'float32': torch.float32, 'fp32': torch.float32, 'bfloat16': torch.bfloat16, 'bf16': torch.bfloat16,
           'float16': torch.float16, 'fp16': torch.float16, 'half': torch.float16,
           torch.float32: torch.float32, torch.bfloat16: torch.bfloat16, torch.float16: torch.float16}


def draw_seed():
    """One 63-bit draw from torch's CPU generator (no device sync): reproducible under torch.manual_seed, different on
    every call, safe from nn.DataParallel's replica threads (the generator locks).  Replaces id()/call counters, which
    repeat on throw-away replicas and are not reproducible."""
    return int(torch.empty((), dtype=torch.int64).random_()) & 0x7FFFFFFFFFFFFFFF


def mix_seed(base, block_index, device):
    """Philox key of one block's dropout for this forward: the per-forward draw, the block's position in the model, the
    rank (ranks seeded alike must not drop the same elements of their different shards) and the device ordinal."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    di = device.index if device.index is not None else 0
    x = (base ^ (0x9E3779B97F4A7C15 * (block_index + 1)) ^ (0xC2B2AE3D27D4EB4F * (rank + 1)) ^ (0x165667B19E3779F9 * (di + 1)))
    x &= 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33
    return x & 0x7FFFFFFFFFFFFFFF


class _ConvHolder(nn.Module):
    """`gcn.branch`: conv + a BatchNorm that upstream declares and never applies (inceptionv2_gcn.py:30,34)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=(1, 1))
        self.bn = nn.BatchNorm2d(cout)


class _GCNParams(nn.Module):
    """Parameter container of the block's GCN unit with the reference's attribute names."""

    def __init__(self, unit, cin, cout, K):
        super().__init__()
        self.kernel_size = K
        if unit == 'incep':
            self.branch = _ConvHolder(cin, cout * K)
        else:
            self.conv = nn.Conv2d(cin, cout * K, kernel_size=(1, 1))

    def the_conv(self):
        return self.branch.conv if hasattr(self, 'branch') else self.conv


def _to_ntvc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _to_nctv(x):
    return x.permute(0, 3, 1, 2)          # logical (N,C,T,V), channels_last strides: no copy


_UNSET = object()          # "argument not given" marker (None is a valid bias term)


class STGCNBlock(nn.Module):
    KIND = None

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dropout=0, residual=True):
        super().__init__()
        var = VARIANTS[self.KIND]
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        self.gcn_kind, self.tcn_kind = var['gcn'], var['tcn']
        self.cin, self.cout, self.stride, self.p_drop = in_channels, out_channels, stride, float(dropout)
        K = kernel_size[1]
        self.K, self.tk = K, kernel_size[0]
        self.gcn = _GCNParams(self.gcn_kind, in_channels, out_channels, K)
        c = out_channels
        if self.tcn_kind == 'single':
            pad = ((kernel_size[0] - 1) // 2, 0)
            self.tcn = nn.Sequential(nn.BatchNorm2d(c), nn.ReLU(inplace=True),
                                     nn.Conv2d(c, c, (kernel_size[0], 1), (stride, 1), pad),
                                     nn.BatchNorm2d(c), nn.Dropout(dropout, inplace=True))
            # the weight-gradient kernel produces [tap][Cout][Cin]; a flat-buffer optimizer may store this parameter (and its
            # gradient) tap-major so that autograd hands the gradient over without a layout copy (harness.FlatSGD)
            self.tcn[2].weight._istgcn_flat_layout = 'tap_major'
        else:
            w = int(c ** 0.5) if self.tcn_kind == 'bneck' else c
            self.width = w
            self.tcn_start = nn.Sequential(nn.BatchNorm2d(c), nn.ReLU(inplace=True))
            if self.tcn_kind == 'bneck':
                self.conv_1x1_start = nn.Conv2d(c, w, (1, 1), (1, 1), (0, 0))
            self.tcn_1 = nn.Conv2d(w, w, (3, 1), (stride, 1), (1, 0))
            self.tcn_2 = nn.Conv2d(w, w, (9, 1), (stride, 1), (4, 0))
            self.tcn_3 = nn.Conv2d(w, w, (15, 1), (stride, 1), (7, 0))
            if self.tcn_kind == 'bneck':
                self.conv_1x1_end = nn.Conv2d(w, c, (1, 1), (1, 1), (0, 0))
            self.tcn_end = nn.Sequential(nn.BatchNorm2d(c), nn.Dropout(dropout, inplace=True))
        if var['dead_linear']:
            self.linear = nn.Linear(3, c)
        if not residual:
            self.res_mode = 'none'
        elif in_channels == out_channels and stride == 1:
            self.res_mode = 'id'
        else:
            self.res_mode = 'conv'
            self.residual = nn.Sequential(nn.Conv2d(in_channels, c, kernel_size=1, stride=(stride, 1)),
                                          nn.BatchNorm2d(c))
        self.relu = nn.ReLU(inplace=True)
        self.block_index = 0               # position in the Model (set by STGCNModel): part of the dropout key

    def batchnorms(self):
        if self.tcn_kind == 'single':
            bns = (self.tcn[0], self.tcn[3])
        else:
            bns = (self.tcn_start[0], self.tcn_end[0])
        return bns + ((self.residual[1],) if self.res_mode == 'conv' else ())

    # ---- inference entry (eval mode, no autograd): folded BatchNorms, cached packed weights --------------
    def infer(self, x, fold_fn, key_extra=(), mst=None, nnz_cap=None):
        """x NTVC -> block output, eval-mode semantics, nothing saved for a backward pass.  `fold_fn()` -> (A_eff, bterm)
        is only called when the cached plan is stale: the plan (folded + fragment-packed weights, Fn.build_infer_plan)
        is keyed on the data pointers and version counters of every tensor it was built from AND on ops.weights_epoch(),
        which every writer that bypasses the version counters bumps (the one-launch optimizer update, the running-
        statistics update of a training forward, a hipGraph replay): load_state_dict / an optimiser step / `.to()` /
        a replayed step rebuild it, and a steady-state eval pass launches only the block's 2-4 kernels."""
        tensors = [t for t in list(self.parameters()) + list(self.buffers()) if t is not None] + list(key_extra)
        if mst is not None:
            tensors.append(mst)
        key = (x.dtype, x.device, nnz_cap, ops.weights_epoch()) + tuple((t.data_ptr(), t._version) for t in tensors)
        cache = self.__dict__.get('_infer_cache')
        if cache is None or cache[0] != key:
            A_eff, bterm = fold_fn()
            cfg, args = self._gather(x, A_eff, mst, nnz_cap, bterm, None)
            (A_e, bt_, Wg3, g1, b1, Wt, bt, g2, b2, Wr, br, gr, betar, Ws, bs, We, be), bns = args
            bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
            plan = Fn.build_infer_plan(cfg, x.dtype, A_e, bt_, Wg3, bn(bns[0]), Wt, bt, bn(bns[1]), Wr, br,
                                       bn(bns[2]) if len(bns) > 2 else None, Ws, bs, We, be)
            cache = (key, cfg, plan)
            self.__dict__['_infer_cache'] = cache
        return Fn.run_infer_plan(cache[1], cache[2], x)

    # ---- engine entry: NTVC in, NTVC out -------------------------------------------------------
    def run(self, x, A_eff, mst=None, nnz_cap=None, bterm=_UNSET, pattern=None, seed_base=None, bump=True, packed=None,
            seed_epoch=None, step_arena=None):
        """pattern: [K,V,V] fp32 sparsity pattern of the adjacency gradient (None = dense); seed_base: the Model's
        per-forward draw (None: drawn here); bump=False: the caller advances num_batches_tracked itself; packed: this
        block's fragment-packed weights from the Model's one-launch PackPlan (None: packed per call); seed_epoch: int64[1]
        device counter added to the dropout seed when the kernels run (Model.device_seed_epoch, for hipGraph replay)."""
        cfg, ((A_eff, bterm, Wg3, g1, b1, Wt, bt, g2, b2, Wr, br, gr, betar, Ws, bs, We, be), bns) = \
            self._gather(x, A_eff, mst, nnz_cap, bterm, pattern)
        cfg.packed = packed
        cfg.seed_epoch = seed_epoch
        cfg.step_arena = step_arena       # ops.StepArena: one zero fill per step for all blocks' gradient accumulators
        bn1, bn2 = bns[0], bns[1]
        bufs = {'bn1': (bn1.running_mean, bn1.running_var), 'bn2': (bn2.running_mean, bn2.running_var)}
        if len(bns) > 2:
            bufs['bnr'] = (bns[2].running_mean, bns[2].running_var)
        training = self.training
        if training and bump:
            torch._foreach_add_([bn.num_batches_tracked for bn in self.batchnorms()], 1)
        seed = 0
        if training and self.p_drop > 0:
            seed = mix_seed(draw_seed() if seed_base is None else seed_base, self.block_index, x.device)
        return Fn.STGCNBlockFn.apply(cfg, training, seed, bufs, x, A_eff, bterm, Wg3, g1, b1, Wt, bt, g2, b2,
                                     Wr, br, gr, betar, Ws, bs, We, be)

    def pack_jobs(self, plan, V):
        """Register this block's weight packs with a PackPlan -> dict of the destination tensors (functional.STGCNBlockFn
        looks them up by name).  Temporal taps that are themselves computed per step (the pre-summed Inception-TCN,
        net/st_gcn_multi3_fix_3A_mstcn.py:212-215) are not parameters and stay with the per-call packer."""
        c, cin, K, s = self.cout, self.cin, self.K, self.stride
        conv = self.gcn.the_conv()
        Wg3 = conv.weight.view(K, c, cin)
        d = {'wg': plan.add_gcn(Wg3.permute(1, 0, 2)), 'wb': plan.add_gcn_wb(Wg3)}
        if self.tcn_kind == 'single':
            Wt = self.tcn[2].weight.view(c, c, self.tk).permute(2, 0, 1)
            taps, in_mul = ops.conv_taps_fwd(self.tk, s)
            d['wt'] = plan.add_tconv(Wt, V, taps, in_mul)
            d['wt_bwd'] = {ph: plan.add_tconv(Wt.transpose(1, 2), V, offs, 1, tap_sel=sel)
                           for ph, offs, sel in Fn.conv_bwd_phases(self.tk, s)}
        if self.res_mode == 'conv':
            Wr = self.residual[0].weight.view(c, cin)
            d['wr'] = plan.add_tconv(Wr.view(1, c, cin), V, [0], s)
            d['wrt'] = plan.add_gcn(Wr.t().unsqueeze(1))
            d['wrg'] = plan.add_gcn(Wr.unsqueeze(1))          # forward form of the same kernel (16-bit storage)
        return d

    def _gather(self, x, A_eff, mst, nnz_cap, bterm, pattern):
        """-> (BlockCfg, ((A_eff, bterm, Wg3, gamma1, beta1, Wt, bt, gamma2, beta2, Wr, br, gamma_r, beta_r, Ws, bs, We, be),
        (bn1, bn2[, bn_residual]))): the block's parameters as the (views of) tensors the kernels take."""
        if not x.is_cuda:
            raise RuntimeError('istgcn_amd: the st_gcn block runs on MI355X only (tensor on %s); no CPU fallback'
                               % x.device)
        c, V = self.cout, x.shape[2]
        conv = self.gcn.the_conv()
        Wg3 = conv.weight.view(self.K, c, self.cin)
        if bterm is _UNSET:                  # (the model passes the fused fold's bias term; standalone blocks fold here)
            bterm = Fn.fold_bias_term(conv.bias, A_eff, c) if conv.bias is not None else None
        if self.tcn_kind == 'single':
            bn1, tconv, bn2 = self.tcn[0], self.tcn[2], self.tcn[3]
            # [k][Cout][Cin] VIEW of the Conv2d weight: the packers read the parameter in place (no copy, no extra launch)
            Wt, bt, ks = tconv.weight.view(c, c, self.tk).permute(2, 0, 1), tconv.bias, self.tk
            Ws = bs = We = be = None
            mode = 'conv'
        else:
            bn1, bn2 = self.tcn_start[0], self.tcn_end[0]
            if mst is None:
                raise TypeError('this st_gcn variant needs mstcn_importance')
            Wt, bt = Fn.fold_tcn_taps_any(self.tcn_1.weight, self.tcn_2.weight, self.tcn_3.weight, self.tcn_1.bias,
                                      self.tcn_2.bias, self.tcn_3.bias, mst,
                                      scale=(1.0 / 3.0) if self.tcn_kind == 'multi3' else 1.0)
            ks = 15
            if self.tcn_kind == 'bneck':
                Ws, bs = self.conv_1x1_start.weight.view(self.width, c), self.conv_1x1_start.bias
                We, be = self.conv_1x1_end.weight.view(c, self.width), self.conv_1x1_end.bias
                mode = 'bneck'
            else:
                Ws = bs = We = be = None
                mode = 'conv'
        bns = (bn1, bn2)
        Wr = br = gr = betar = None
        mom, eps = bn1.momentum, bn1.eps
        if self.res_mode == 'conv':
            rc, rbn = self.residual[0], self.residual[1]
            Wr, br, gr, betar = rc.weight.view(c, self.cin), rc.bias, rbn.weight, rbn.bias
            bns = (bn1, bn2, rbn)
        if nnz_cap is None:
            nnz_cap = self.K * V * V
        cfg = Fn.BlockCfg(self.cin, c, self.K, V, self.stride, self.res_mode, mode, ks, self.p_drop, int(nnz_cap),
                          width=getattr(self, 'width', None), momentum=mom if mom is not None else 0.1, eps=eps,
                          pattern=pattern)
        return cfg, ((A_eff, bterm, Wg3, bn1.weight, bn1.bias, Wt, bt, bn2.weight, bn2.bias, Wr, br, gr, betar,
                      Ws, bs, We, be), bns)

    # ---- reference-signature forward on (N,C,T,V) tensors ------------------------------------------
    def forward(self, x, A, *rest):
        gk, multi = self.gcn_kind, self.tcn_kind != 'single'
        mst = None
        if gk == 'plain':
            assert A.size(0) == self.K
            if multi:
                (mst,) = rest
            A_eff, ret = A, (A,)
        elif gk == 'incep':
            A2, A3 = rest[0], rest[1]
            assert A.size(0) == self.K
            A_eff, ret = A + A2 + A3, (A, A2, A3)
        else:
            i1, i2, i3 = rest[0], rest[1], rest[2]
            assert A.size(0) == self.K
            if multi:
                mst = rest[3]
            A_eff, ret = Fn.fold_adjacency('3a', A, (i1, i2, i3)), (A,)
        y = self.run(_to_ntvc(x), A_eff, mst)
        return (_to_nctv(y),) + ret


def _buffers_reloaded(module, incompatible_keys):
    """load_state_dict may have replaced the adjacency buffers: drop everything derived from them."""
    module._nnz_cap = None
    module._fold_consts.clear()
    module._patterns.clear()


_PLAN_LOCK = threading.Lock()


class STGCNModel(nn.Module):
    KIND = None
    BLOCK = None

    def __init__(self, in_channels, num_class, graph_args, edge_importance_weighting, **kwargs):
        super().__init__()
        var = VARIANTS[self.KIND]
        self.gcn_kind, self.tcn_kind = var['gcn'], var['tcn']
        self.act_dtype = _DTYPES[kwargs.pop('compute_dtype', 'float32')]
        # optional GPU feeder stage: dict(window_size=, random_choose=, random_move=) = the options of
        # feeder.Feeder (feeder/feeder.py:33-45), applied to the raw clips inside the data_bn prologue while training
        ga = kwargs.pop('gpu_augment', None)
        self.gpu_augment = None
        if ga:
            from ..feeder_gpu import GpuAugment
            self.gpu_augment = GpuAugment(**ga)
        self.graph = Graph(**graph_args)
        if self.gcn_kind == 'incep':                                    # st_gcn_msgcn.py:36-39
            self.register_buffer('A2', torch.tensor(self.graph.A2, dtype=torch.float32, requires_grad=False))
            self.register_buffer('A3', torch.tensor(self.graph.A3, dtype=torch.float32, requires_grad=False))
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer('A', A)
        K, V = A.size(0), A.size(1)
        kernel_size = (9, K)
        self.data_bn = nn.BatchNorm1d(in_channels * V)
        kwargs0 = {k: v for k, v in kwargs.items() if k != 'dropout'}
        blocks, cin = [], in_channels
        for idx, (cout, stride) in enumerate(var['plan']):
            if idx == 0:
                blocks.append(self.BLOCK(cin, cout, kernel_size, 1, residual=False, **kwargs0))
            else:
                blocks.append(self.BLOCK(cin, cout, kernel_size, stride, **kwargs))
            blocks[-1].block_index = idx
            cin = cout
        self.st_gcn_networks = nn.ModuleList(blocks)
        n_imp = 3 if self.gcn_kind in ('incep', '3a') else 1
        shapes = [self.A.size(), self.A2.size() if self.gcn_kind == 'incep' else self.A.size(),
                  self.A3.size() if self.gcn_kind == 'incep' else self.A.size()]
        for j, name in enumerate(('edge_importance', 'edge_importance2', 'edge_importance3')[:n_imp]):
            if edge_importance_weighting:
                setattr(self, name, nn.ParameterList([nn.Parameter(torch.ones(shapes[j])) for _ in blocks]))
            else:
                setattr(self, name, [1] * len(blocks))
        if self.tcn_kind != 'single':
            self.mstcn_importance = nn.ParameterList([nn.Parameter(torch.ones(3)) for _ in blocks])
        self.fcn = nn.Conv2d(256, num_class, kernel_size=1)
        self._nnz_cap = None
        self._fold_consts = {}               # device -> [J,K,V,V] constants of the fused importance fold
        self._patterns = {}                  # device -> [K,V,V] fp32 union pattern of those constants
        self._pack_plans = {}                # (dtype, device, V) -> weight-pack plan; the dict is shared with DataParallel replicas
        self.register_load_state_dict_post_hook(_buffers_reloaded)

    # the sparsity pattern bounds the kernels' in-LDS adjacency lists; recomputed if buffers are reloaded
    def _cap(self):
        if self._nnz_cap is None:
            pat = self.A != 0
            if self.gcn_kind == 'incep':
                pat = pat | (self.A2 != 0) | (self.A3 != 0)
            self._nnz_cap = max(1, int(pat.sum().item()))
        return self._nnz_cap

    def _a_eff(self, i):
        imps = [self.edge_importance[i]]
        if self.gcn_kind in ('incep', '3a'):
            imps += [self.edge_importance2[i], self.edge_importance3[i]]
        return Fn.fold_adjacency(self.gcn_kind, self.A, imps, getattr(self, 'A2', None), getattr(self, 'A3', None))

    def _fold_B(self, dev):
        """[J,K,V,V] constants B_j with A_eff = sum_j B_j (.) importance_j (cached per device until buffers reload)."""
        B = self._fold_consts.get(dev)
        if B is None:
            if self.gcn_kind == 'plain':
                mats = [self.A]
            elif self.gcn_kind == 'incep':
                mats = [self.A, self.A2, self.A3]
            else:
                mats = [self.A, self.A * self.A, self.A * self.A * self.A]
            B = self._fold_consts[dev] = torch.stack([m.to(dev) for m in mats]).contiguous()
        return B

    def _pattern(self, dev):
        """[K,V,V] fp32, 1 where any constant adjacency of the fold is non-zero: the entries whose importance gradient
        exists.  Fixed by the buffers, so an importance value that reaches exactly 0 keeps receiving its gradient."""
        pat = self._patterns.get(dev)
        if pat is None:
            pat = self._patterns[dev] = (self._fold_B(dev) != 0).any(0).float().contiguous()
        return pat

    def _folded(self, i, blk):
        """(A_eff, bterm-or-_UNSET) of block i: one fused launch when the importances are learnable parameters."""
        imps = [self.edge_importance[i]]
        if self.gcn_kind in ('incep', '3a'):
            imps += [self.edge_importance2[i], self.edge_importance3[i]]
        K, V = self.A.shape[0], self.A.shape[1]
        if all(isinstance(p, torch.Tensor) and p.is_cuda for p in imps) and K * V <= 512 and K * V * V <= 12288:
            conv = blk.gcn.the_conv()
            return Fn.FoldFn.apply(self._fold_B(imps[0].device), conv.bias, blk.cout, *imps)
        return self._a_eff(i), _UNSET

    def _folded_all(self):
        """[(A_eff_i, bterm_i)] of every block from ONE launch (Fn.FoldAllFn), or None when the fused fold does not apply
        (importances that are not GPU parameters, too many blocks / joints): the caller then folds per block."""
        blocks = list(self.st_gcn_networks)
        names = ['edge_importance'] + (['edge_importance2', 'edge_importance3'] if self.gcn_kind in ('incep', '3a') else [])
        imps = [[getattr(self, n)[i] for n in names] for i in range(len(blocks))]
        K, V = self.A.shape[0], self.A.shape[1]
        ok = all(isinstance(p, torch.Tensor) and p.is_cuda for imp in imps for p in imp)
        if not ok or len(blocks) > 16 or K * V > 512 or K * V * V > 12288:
            return None
        biases = [blk.gcn.the_conv().bias for blk in blocks]
        Cs = tuple(blk.cout for blk in blocks)
        out = Fn.FoldAllFn.apply(self._fold_B(imps[0][0].device), Cs, len(names), *biases, *[p for imp in imps for p in imp])
        return [(out[2 * i], out[2 * i + 1]) for i in range(len(blocks))]

    def device_seed_epoch(self, device=None):
        """Switch the dropout masks to `seed + *epoch` with `epoch` an int64[1] counter on the GPU (returned; created on
        first use): a training step recorded once in a hipGraph then draws fresh masks on every replay as long as the
        counter is advanced once per step (harness.GraphedStep records `epoch += 1` at the head of the graph).  The host
        seed is still drawn per eager forward; under replay it is the value drawn at capture time."""
        ep = self.__dict__.get('_seed_epoch')
        if ep is None:
            device = device if device is not None else next(self.parameters()).device
            ep = torch.zeros(1, dtype=torch.int64, device=device)
            self.__dict__['_seed_epoch'] = ep
        return ep

    def _packed_weights(self, x):
        """Every weight pack of the trunk in ONE launch per forward (they were ~46 launches per step): the plan is rebuilt
        when a parameter's storage moved (an optimizer re-pointing `.data` into its flat buffer, `.to()`), the launch is
        repeated every forward because the optimizer has updated the weights in between."""
        ptrs = tuple((p.data_ptr(), p.stride()) for p in self.parameters())
        # One plan per (storage type, device, joints), kept in a dict that nn.DataParallel's replicas SHARE with the
        # module they were made from (a replica's __dict__ is a shallow copy: same dict object): a replica -- a fresh set of
        # parameter tensors on every forward -- finds its device's plan, re-points the job sources and reuses the
        # destinations, instead of building a plan (geometry queries, ~46 allocations) per forward and device.
        plans = self.__dict__.get('_pack_plans')
        if plans is None:
            plans = self.__dict__['_pack_plans'] = {}
        key = (x.dtype, x.device, x.shape[2])
        with _PLAN_LOCK:                                    # (replicas run in one host thread each)
            ent = plans.get(key)
            if ent is None:
                plan = ops.PackPlan(x.dtype, x.device)
                ent = plans[key] = [ptrs, plan, [blk.pack_jobs(plan, x.shape[2]) for blk in self.st_gcn_networks], threading.Lock()]
        with ent[3]:                                        # two replicas on ONE device share a plan: re-point + launch as one step
            if ent[0] != ptrs:
                ent[1].begin_rebind()
                ent[2] = [blk.pack_jobs(ent[1], x.shape[2]) for blk in self.st_gcn_networks]
                ent[1].end_rebind()
                ent[0] = ptrs
            ent[1].run()
            return ent[2]

    def _folded_bias(self, i, blk):
        """(A_eff, bterm) with the bias term resolved (for the inference plan)."""
        A_eff, bterm = self._folded(i, blk)
        if bterm is _UNSET:
            conv = blk.gcn.the_conv()
            bterm = Fn.fold_bias_term(conv.bias, A_eff, blk.cout) if conv.bias is not None else None
        return A_eff, bterm

    def _trunk(self, x):
        if not x.is_cuda:
            raise RuntimeError('istgcn_amd.net: Model.forward needs the input on an MI355X (got %s); the HIP path has '
                               'no CPU fallback' % x.device)
        N, C, T, V, M = x.size()
        # input stage (st_gcnold.py:74-80): (N,C,T,V,M) -> BatchNorm1d over the V*C channels -> NTVC activation, with the
        # feeder's augmentation (feeder/tools.py:31-101) folded in when configured: csrc/input.hip, 2 launches
        bn = self.data_bn
        shift = move = Tw = None
        if self.gpu_augment is not None and self.training:
            shift, move, Tw = self.gpu_augment.draw(N, T)
            shift = None if shift is None else shift.to(x.device, non_blocking=True)
            move = None if move is None else move.to(x.device, non_blocking=True)
        x = Fn.InputStageFn.apply(x.float(), bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                  bn.momentum if bn.momentum is not None else 0.1, bn.eps, self.act_dtype, shift, move, Tw)
        cap = self._cap()
        pat = self._pattern(x.device)
        seed_base = None
        if self.training:
            seed_base = draw_seed()
            # every BatchNorm of the model advances its counter: ONE multi-tensor launch instead of 2-3 per block
            torch._foreach_add_([bn.num_batches_tracked] + [b.num_batches_tracked for blk in self.st_gcn_networks
                                                            for b in blk.batchnorms()], 1)
        infer = (not self.training) and (not torch.is_grad_enabled())
        packed = None if infer else self._packed_weights(x)
        folds = None if infer else self._folded_all()
        arena = None
        if self.training and torch.is_grad_enabled():
            # gradient accumulators of all blocks out of one zero-filled allocation, sized by what the last backward took
            prev = self.__dict__.get('_step_arena')
            hint = max(self.__dict__.get('_arena_hint', 0), prev.requested if prev is not None else 0)
            self.__dict__['_arena_hint'] = hint
            arena = self.__dict__['_step_arena'] = ops.StepArena(hint)
        for i, blk in enumerate(self.st_gcn_networks):
            mst = self.mstcn_importance[i] if self.tcn_kind != 'single' else None
            if infer:                      # folded BatchNorms, cached plans, 2-4 launches per block
                imps = [getattr(self, n)[i] for n in ('edge_importance', 'edge_importance2', 'edge_importance3')
                        if hasattr(self, n)]
                x = blk.infer(x, lambda i=i, blk=blk: self._folded_bias(i, blk),
                              key_extra=[t for t in imps if isinstance(t, torch.Tensor)] + [self.A], mst=mst, nnz_cap=cap)
                continue
            A_eff, bterm = folds[i] if folds is not None else self._folded(i, blk)
            x = blk.run(x, A_eff, mst, nnz_cap=cap, bterm=bterm, pattern=pat, seed_base=seed_base, bump=False,
                        packed=packed[i], seed_epoch=self.__dict__.get('_seed_epoch'), step_arena=arena)
        return x

    def forward(self, x):
        N, M = x.size(0), x.size(4)
        y = self._trunk(x)                                           # (NM, T', V, 256)
        # global pooling over (T, V) and the persons, fp32 accumulation straight from the storage type (csrc/pointwise.hip)
        feat = Fn.PoolFn.apply(y, M)
        # fcn is a 1x1 Conv2d on a 1x1 map (st_gcnold.py:92-94) = a matrix product (a plain GEMM instead of a convolution
        # library's fallback kernels)
        return F.linear(feat, self.fcn.weight.view(self.fcn.weight.shape[0], -1), self.fcn.bias)

    def extract_feature(self, x):
        N, M = x.size(0), x.size(4)
        y = self._trunk(x).float()
        _, t, v, c = y.size()
        feature = y.view(N, M, t, v, c).permute(0, 4, 2, 3, 1)
        o = F.conv2d(y.permute(0, 3, 1, 2), self.fcn.weight, self.fcn.bias)      # (NM, nc, t, v)
        output = o.view(N, M, -1, t, v).permute(0, 2, 3, 4, 1)
        return output, feature


def make(kind):
    """-> (Model, st_gcn) classes of one reference model file."""
    blk = type('st_gcn', (STGCNBlock,), {'KIND': kind, '__doc__': 'st_gcn block of net/%s.py' % kind})
    mdl = type('Model', (STGCNModel,), {'KIND': kind, 'BLOCK': blk, '__doc__': 'Model of net/%s.py' % kind})
    return mdl, blk
