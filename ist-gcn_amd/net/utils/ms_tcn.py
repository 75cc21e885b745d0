"""net/utils/ms_tcn.py:5-52 on the HIP kernels.  Upstream this class is dead code: the only model that instantiates it
(net/st_gcn_mstgcn.py:214) calls it with the wrong arity and crashes, and the models that import it never build it
(SURVEY.md 2.1 #5).  It is provided for import compatibility with identical parameters / state_dict keys, and its forward
(:41-52: BatchNorm -> ReLU -> conv_b -> the SAME BatchNorm -> Dropout; conv_a, conv_c and `mstcn_importance` unused) runs on
the library's kernels: the first BatchNorm + ReLU are applied inside the temporal conv's staging, the second BatchNorm's
batch sums come out of its epilogue.  Forward only (the result carries no autograd graph): nothing upstream trains it."""
import torch
import torch.nn as nn

from ... import ops


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

    @torch.no_grad()
    def forward(self, x, mstcn_importance=None):
        """x: (N, C, T, V) as the reference passes it -> (N, C, T/stride, V)."""
        bn, conv = self.batchnorm2d, self.conv_b
        N, C, T, V = x.shape
        s, k = self.stride, conv.kernel_size[0]
        if k % 2 == 0:
            raise ValueError('MSTCN: even temporal kernel sizes are not supported (padding (k-1)//2 is asymmetric)')
        xs = x.permute(0, 2, 3, 1).contiguous()                       # NTVC (istgcn.h)
        dt = xs.dtype
        training = self.training
        mom = bn.momentum if bn.momentum is not None else 0.1
        # first BatchNorm (ms_tcn.py:42): its batch sums -- the input is not produced by one of the library's kernels, so
        # the two sums are taken here -- then istgcn_bn_finalize (coefficients + running statistics, as nn.BatchNorm2d)
        st1 = None
        if training:
            xd = xs.reshape(-1, C).double()
            st1 = torch.zeros((ops.STATS_REP, 2, C), dtype=torch.float64, device=xs.device)
            st1[0, 0], st1[0, 1] = xd.sum(0), (xd * xd).sum(0)
        coef1 = ops.bn_finalize(st1, N * T * V, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, training)
        # ReLU + conv_b (:43,45) with the BatchNorm affine applied on the way in; epilogue: batch sums of the output
        taps, in_mul = ops.conv_taps_fwd(k, s)
        Tz = (T - 1) // s + 1
        wt = ops.pack_tconv_weight(conv.weight.view(C, C, k).permute(2, 0, 1), V, taps, in_mul, dt)
        st2 = ops.new_stats(C, xs.device) if training else None
        z = ops.tconv(xs, wt, C, taps, bias=conv.bias, pre=coef1[:2].contiguous(), pre_relu=True, stats=st2,
                      Tout=Tz, Mlog=Tz, in_mul=in_mul)
        # the SAME BatchNorm again (:50): second batch statistics, second running-statistics update
        coef2 = ops.bn_finalize(st2, N * Tz * V, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, training)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 2
        abc = torch.stack([coef2[0], torch.zeros_like(coef2[0]), coef2[1]]).contiguous()
        y = ops.affine2(z, z, abc)                                    # y = scale * z + 0 * z + shift
        y = y.permute(0, 3, 1, 2)
        p = self.dropout.p if training else 0.0
        if p > 0:                                                      # :51 (torch's generator: the masks cannot match anyway)
            y = torch.nn.functional.dropout(y, p, True)
        return y
