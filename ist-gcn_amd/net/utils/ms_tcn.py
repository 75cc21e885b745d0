"""Import-compat for net/utils/ms_tcn.py:5-52.  Upstream this class is dead code: the only model that instantiates
it (net/st_gcn_mstgcn.py:214) calls it with the wrong arity and crashes, and the models that import it never build
it.  The parameters are declared so checkpoints/keys line up; calling it raises."""
import torch.nn as nn


class MSTCN(nn.Module):
    def __init__(self, out_channels, kernel_size_a, kernel_size_b, kernel_size_c, dropout, stride=1):
        super().__init__()
        c = out_channels
        self.batchnorm2d = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        for name, k in (('conv_a', kernel_size_a), ('conv_b', kernel_size_b), ('conv_c', kernel_size_c)):
            setattr(self, name, nn.Conv2d(c, c, kernel_size=(k, 1), stride=(stride, 1), padding=((k - 1) // 2, 0)))
        self.dropout = nn.Dropout(dropout, inplace=True)

    def forward(self, x, mstcn_importance):
        raise NotImplementedError('MSTCN is unreachable in every working reference model; the Inception-TCN path is '
                                  'net.st_gcn_mstcn / st_gcn_mstcn_1x1 / st_gcn_multi3_fix_3A_mstcn')
