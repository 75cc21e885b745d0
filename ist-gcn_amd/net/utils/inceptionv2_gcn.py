"""Inception-GCN unit drop-in (reference: net/utils/inceptionv2_gcn.py:7-89, inceptionv2_gcn_new.py:7-62): one shared
1x1 conv, einsum against A, A2, A3, summed == one graph conv against A + A2 + A3."""
import torch.nn as nn

from ... import functional as Fn


class BasicConv2d(nn.Module):
    """conv + a BatchNorm that is declared but never applied upstream (inceptionv2_gcn.py:30-35)."""

    def __init__(self, in_channels, out_channels, kernel_size, t_padding=0, t_kernel_size=1, t_stride=1,
                 t_dilation=1, bias=True):
        super().__init__()
        if (t_kernel_size, t_stride, t_padding, t_dilation) != (1, 1, 0, 1):
            raise NotImplementedError('only the 1x1 channel expansion the reference models use is supported')
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(1, 1), bias=bias)
        self.bn = nn.BatchNorm2d(out_channels * kernel_size)


class Inception2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, t_padding=0, t_kernel_size=1, t_stride=1,
                 t_dilation=1, bias=True):
        super().__init__()
        self.kernel_size = kernel_size
        self.out_channels = out_channels
        self.branch = BasicConv2d(in_channels, out_channels, kernel_size, t_padding, t_kernel_size, t_stride,
                                  t_dilation, bias)

    def forward(self, x, A, A2, A3):
        assert A.size(0) == self.kernel_size
        if not x.is_cuda:
            raise RuntimeError('istgcn_amd: graph convolution runs on MI355X only; no CPU fallback')
        conv, c = self.branch.conv, self.out_channels
        A_eff = A + A2 + A3
        W3 = conv.weight.view(self.kernel_size, c, -1)
        bterm = Fn.fold_bias_term(conv.bias, A_eff, c) if conv.bias is not None else None
        y = Fn.GraphConvFn.apply(x.permute(0, 2, 3, 1).contiguous(), A_eff, bterm, W3, A_eff.numel())
        return y.permute(0, 3, 1, 2), A, A2, A3
