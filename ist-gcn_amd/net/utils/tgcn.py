"""GCN-unit drop-ins (reference: net/utils/tgcn.py:7-89 and its one-line variants tgcn_multi3.py, tgcn_multi3_fix.py,
tgcn_only3.py, tgcn_multi3_fix_3A.py:76-92).  All of them are the same HIP kernel with a different effective
adjacency; `forward` takes / returns (N, C, T, V) tensors like upstream."""
import torch.nn as nn

from ... import functional as Fn


class _GraphConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1, t_padding=0,
                 t_dilation=1, bias=True):
        super().__init__()
        if (t_kernel_size, t_stride, t_padding, t_dilation) != (1, 1, 0, 1):
            raise NotImplementedError('only the 1x1 channel expansion the reference models use is supported')
        self.kernel_size = kernel_size
        self.out_channels = out_channels
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(1, 1), bias=bias)

    def _apply_graph(self, x, A_eff):
        if not x.is_cuda:
            raise RuntimeError('istgcn_amd: graph convolution runs on MI355X only; no CPU fallback')
        c = self.out_channels
        W3 = self.conv.weight.view(self.kernel_size, c, -1)
        bterm = Fn.fold_bias_term(self.conv.bias, A_eff, c) if self.conv.bias is not None else None
        y = Fn.GraphConvFn.apply(x.permute(0, 2, 3, 1).contiguous(), A_eff, bterm, W3, A_eff.numel())
        return y.permute(0, 3, 1, 2)


class ConvTemporalGraphical(_GraphConvBase):
    """net/utils/tgcn.py:76-89"""

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        return self._apply_graph(x, A), A


class ConvTemporalGraphical3A(_GraphConvBase):
    """net/utils/tgcn_multi3_fix_3A.py:76-92"""

    def forward(self, x, A, importance, importance2, importance3):
        assert A.size(0) == self.kernel_size
        return self._apply_graph(x, Fn.fold_adjacency('3a', A, (importance, importance2, importance3))), A


class ConvTemporalGraphicalMulti3(_GraphConvBase):
    """net/utils/tgcn_multi3.py:86-89  (A + A**2 + A**3), `average=True`: tgcn_multi3_fix.py:89 (/3)"""
    average = False

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        Ae = A + A * A + A * A * A
        return self._apply_graph(x, Ae / 3 if self.average else Ae), A


class ConvTemporalGraphicalMulti3Fix(ConvTemporalGraphicalMulti3):
    average = True


class ConvTemporalGraphicalOnly3(_GraphConvBase):
    """net/utils/tgcn_only3.py:86"""

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        return self._apply_graph(x, A * A * A), A
