"""net/utils/ms_tcn.py:5-52 on the HIP kernels.  Upstream this class is dead code: the only model that instantiates it
(net/st_gcn_mstgcn.py:214) calls it with the wrong arity and crashes, and the models that import it never build it
(SURVEY.md 2.1 #5).  It is provided for import compatibility with identical parameters / state_dict keys, and its forward
(:41-52: BatchNorm -> ReLU -> conv_b -> the SAME BatchNorm -> Dropout; conv_a, conv_c and `mstcn_importance` unused) runs on
the library's kernels: the first BatchNorm + ReLU are applied inside the temporal conv's staging, the second BatchNorm's
batch sums come out of its epilogue.  Backward (round 5; autograd of the reference's forward is the definition): the second
BatchNorm's elementwise half (`affine2`), the temporal conv's weight gradient and its data gradient with the ReLU mask and
the first BatchNorm's backward sums fused, `affine2` again; the ONE BatchNorm's parameters collect both applications'
gradients.  conv_a / conv_c get no gradient (None), as upstream."""
import torch
import torch.nn as nn

from ... import ops
from ...functional import _conv_bwd_data


def _sums64(a, b=None):
    """[STATS_REP][2][C] fp64 with the per-channel sums of a (and of a*b, else of a*a) in row 0 -- for tensors no kernel
    of the library produced (the class's input, the incoming gradient)."""
    C = a.shape[-1]
    ad = a.reshape(-1, C).double()
    st = torch.zeros((ops.STATS_REP, 2, C), dtype=torch.float64, device=a.device)
    st[0, 0] = ad.sum(0)
    st[0, 1] = (ad * (ad if b is None else b.reshape(-1, C).double())).sum(0)
    return st


class MSTCNFn(torch.autograd.Function):
    """xs [N,T,V,C] (NTVC, istgcn.h) -> BatchNorm(conv_b(ReLU(BatchNorm(xs)))) [N,Tz,V,C]; gamma / beta / running statistics
    are the ONE BatchNorm's, W [C][C][k][1] and b conv_b's."""

    @staticmethod
    def forward(ctx, xs, gamma, beta, W, b, bn, k, s, training):
        N, T, V, C = xs.shape
        dt = xs.dtype
        mom = bn.momentum if bn.momentum is not None else 0.1
        # first BatchNorm (ms_tcn.py:42): its batch sums -- the input is not produced by one of the library's kernels, so
        # the two sums are taken here -- then istgcn_bn_finalize (coefficients + running statistics, as nn.BatchNorm2d)
        st1 = _sums64(xs) if training else None
        coef1 = ops.bn_finalize(st1, N * T * V, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, training)
        # ReLU + conv_b (:43,45) with the BatchNorm affine applied on the way in; epilogue: batch sums of the output
        taps, in_mul = ops.conv_taps_fwd(k, s)
        Tz = (T - 1) // s + 1
        Wt = W.view(C, C, k).permute(2, 0, 1)                        # [k][Cout][Cin]
        wt = ops.pack_tconv_weight(Wt, V, taps, in_mul, dt)
        st2 = ops.new_stats(C, xs.device) if training else None
        z = ops.tconv(xs, wt, C, taps, bias=b, pre=coef1[:2].contiguous(), pre_relu=True, stats=st2,
                      Tout=Tz, Mlog=Tz, in_mul=in_mul)
        # the SAME BatchNorm again (:50): second batch statistics, second running-statistics update
        coef2 = ops.bn_finalize(st2, N * Tz * V, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, training)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 2
        abc = torch.stack([coef2[0], torch.zeros_like(coef2[0]), coef2[1]]).contiguous()
        y = ops.affine2(z, z, abc)                                    # y = scale * z + 0 * z + shift
        ctx.save_for_backward(xs, z, coef1, coef2, gamma, W)
        ctx.meta = (k, s, training)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, z, coef1, coef2, gamma, W = ctx.saved_tensors
        k, s, training = ctx.meta
        N, T, V, C = xs.shape
        Tz = z.shape[1]
        dy = dy.contiguous()
        # second BatchNorm: sum dy, sum dy * zhat (the incoming gradient is nobody's epilogue: taken here), coefficients, dz
        zhat = (z.reshape(-1, C).double() - coef2[2].double()) * coef2[3].double()
        abc2, dg2, db2 = ops.bn_bwd_coef(_sums64(dy, zhat), N * Tz * V, gamma, coef2, training)
        dz = ops.affine2(dy, z, abc2)
        # conv_b: weight gradient (its input re-staged through BatchNorm + ReLU), data gradient with the ReLU mask and the
        # first BatchNorm's backward sums fused
        taps, in_mul = ops.conv_taps_fwd(k, s)
        pre1 = coef1[:2].contiguous()
        dWt, dbc = ops.tconv_wgrad(dz, xs, taps, in_mul=in_mul, pre=pre1, pre_relu=True, want_bias=True)
        st1b = ops.new_stats(C, xs.device)
        Wt = W.view(C, C, k).permute(2, 0, 1)
        d1 = _conv_bwd_data(dz, Wt, k, s, T, C, V, aux=xs, maux=coef1, stats=st1b)
        abc1, dg1, db1 = ops.bn_bwd_coef(st1b, N * T * V, gamma, coef1, training)
        dx = ops.affine2(d1, xs, abc1)
        return dx, dg1 + dg2, db1 + db2, dWt.permute(1, 2, 0).reshape(W.shape), dbc, None, None, None, None


class MSTCN(nn.Module):
    def __init__(self, out_channels, kernel_size_a, kernel_size_b, kernel_size_c, dropout, stride=1):
        super().__init__()
        c = out_channels
        self.batchnorm2d = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        for name, k in (('conv_a', kernel_size_a), ('conv_b', kernel_size_b), ('conv_c', kernel_size_c)):
            setattr(self, name, nn.Conv2d(c, c, kernel_size=(k, 1), stride=(stride, 1), padding=((k - 1) // 2, 0)))
        self.dropout = nn.Dropout(dropout, inplace=True)
        self.stride = stride

    def forward(self, x, mstcn_importance=None):
        """x: (N, C, T, V) as the reference passes it -> (N, C, T/stride, V)."""
        bn, conv = self.batchnorm2d, self.conv_b
        k = conv.kernel_size[0]
        if k % 2 == 0:
            raise ValueError('MSTCN: even temporal kernel sizes are not supported (padding (k-1)//2 is asymmetric)')
        xs = x.permute(0, 2, 3, 1).contiguous()                       # NTVC (istgcn.h)
        y = MSTCNFn.apply(xs, bn.weight, bn.bias, conv.weight, conv.bias, bn, k, self.stride, self.training)
        y = y.permute(0, 3, 1, 2)
        p = self.dropout.p if self.training else 0.0
        if p > 0:                                                      # :51 (torch's generator: the masks cannot match anyway)
            y = torch.nn.functional.dropout(y, p, True)
        return y
