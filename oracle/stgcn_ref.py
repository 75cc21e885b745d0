"""ORACLE (test infrastructure, never shipped, never measured as the product).

Plain-PyTorch fp32 CPU restatement of the IST-GCN hot path: GCN units, st_gcn blocks and
the Model wrappers of the reference `net/` package.  One table-driven implementation
covers every variant; each piece cites the reference file:line it follows.  Module /
parameter names are the reference's, so `state_dict()` keys and shapes are identical
(pinned by tests/golden/state_dict_g5.json) and the same weights load into the reference,
this oracle and the HIP product.

Pinned by: tests/golden/{units_g2,block_g3_*,model_g4_*}.npz, produced by importing the
reference (tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product (`ist-gcn_amd/`) never does.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .graph_ref import GraphRef

# block plans: (in, out, stride) per st_gcn block; block 0 has no residual and no dropout.
PLAN10 = [(None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1),
          (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 256, 1)]
PLAN7 = [(None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1)]
PLAN13 = [(None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1),
          (128, 128, 1), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 256, 1), (256, 256, 1)]

# kind -> (gcn unit, tcn unit, plan, has the dead nn.Linear(3,C))
#   gcn: 'plain'  net/utils/tgcn.py:76-89            (caller passes A*imp, st_gcnold.py:86)
#        'incep'  net/utils/inceptionv2_gcn.py:64-89 (caller passes A*imp, A2*imp2, A3*imp3, st_gcn_msgcn.py:116-117)
#        '3a'     net/utils/tgcn_multi3_fix_3A.py:76-92 (raw A + 3 importances, elementwise powers)
#   tcn: 'single' st_gcnold.py:164-176   'multi' st_gcn_multi3_fix_3A_mstcn.py:156-184,206-220
#        'multi3' st_gcn_mstcn.py:235-249 (the /3)   'bneck' st_gcn_mstcn_1x1.py:186-228,250-266
KINDS = {
    'st_gcnold': ('plain', 'single', PLAN10, True),
    'st_gcn_tanh': ('plain', 'single', PLAN10, True),
    'st_gcn_msgcn': ('incep', 'single', PLAN10, False),
    'st_gcn_msgcn_new': ('incep', 'single', PLAN7, False),
    'st_gcn_deep_msgcn': ('incep', 'single', PLAN13, False),
    'st_gcn_mstcn': ('plain', 'multi3', PLAN7, False),
    'st_gcn_mstcn_1x1': ('plain', 'bneck', PLAN10, False),
    'st_gcn_mstcn_1x1_deep': ('plain', 'bneck', PLAN13, False),
    'st_gcn_multi3_fix_3A_mstcn': ('3a', 'multi', PLAN10, False),
}


def graph_einsum(h, A):
    """einsum('nkctv,kvw->nctw') of tgcn.py:86 written as one matmul.
    h: (N, K*C, T, V) conv output, A: (K, V, V)."""
    n, kc, t, v = h.shape
    k = A.shape[0]
    c = kc // k
    hk = h.view(n, k, c, t, v).permute(0, 2, 3, 1, 4).reshape(n, c, t, k * v)
    return hk @ A.reshape(k * v, v)


class _Conv(nn.Module):
    """holder so that keys read '<name>.conv.weight' like the reference's wrappers."""

    def __init__(self, cin, cout, bn=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=(1, 1))
        if bn:  # declared, never applied: inceptionv2_gcn.py:30,34
            self.bn = nn.BatchNorm2d(cout)


class RefMSTCN(nn.Module):
    """net/utils/ms_tcn.py:5-52 (dead code upstream): BatchNorm -> ReLU -> conv_b -> the SAME BatchNorm ->
    Dropout; conv_a / conv_c are declared and never used, `mstcn_importance` is ignored (ms_tcn.py:44-49)."""

    def __init__(self, out_channels, kernel_size_a, kernel_size_b, kernel_size_c, dropout, stride=1):
        super().__init__()
        c = out_channels
        self.batchnorm2d = nn.BatchNorm2d(c)
        for name, k in (('conv_a', kernel_size_a), ('conv_b', kernel_size_b), ('conv_c', kernel_size_c)):
            setattr(self, name, nn.Conv2d(c, c, (k, 1), (stride, 1), ((k - 1) // 2, 0)))
        self.p = dropout

    def forward(self, x, mstcn_importance=None):
        x = self.conv_b(F.relu(self.batchnorm2d(x)))             # ms_tcn.py:42-45
        return F.dropout(self.batchnorm2d(x), self.p, self.training)   # :50-51


class RefGCN(nn.Module):
    def __init__(self, unit, cin, cout, K):
        super().__init__()
        self.unit, self.K = unit, K
        if unit == 'incep':
            self.branch = _Conv(cin, cout * K, bn=True)
        else:
            self.conv = nn.Conv2d(cin, cout * K, kernel_size=(1, 1))

    def forward(self, x, adj):
        conv = self.branch.conv if self.unit == 'incep' else self.conv
        h = conv(x)
        if self.unit == 'plain':
            (a,) = adj
            assert a.size(0) == self.K
            return graph_einsum(h, a)
        if self.unit == 'incep':
            a1, a2, a3 = adj
            assert a1.size(0) == self.K
            return graph_einsum(h, a1) + graph_einsum(h, a2) + graph_einsum(h, a3)
        A, i1, i2, i3 = adj
        assert A.size(0) == self.K
        return graph_einsum(h, A * i1) + graph_einsum(h, A ** 2 * i2) + graph_einsum(h, A ** 3 * i3)


class RefBlock(nn.Module):
    def __init__(self, kind, cin, cout, K, stride=1, dropout=0, residual=True):
        super().__init__()
        gcn, tcn, _, dead_linear = KINDS[kind]
        self.tcn_kind, self.stride = tcn, stride
        self.gcn = RefGCN(gcn, cin, cout, K)
        if tcn == 'single':
            self.tcn = nn.Sequential(
                nn.BatchNorm2d(cout), nn.ReLU(inplace=False),
                nn.Conv2d(cout, cout, (9, 1), (stride, 1), (4, 0)),
                nn.BatchNorm2d(cout), nn.Dropout(dropout))
        else:
            width = int(cout ** 0.5) if tcn == 'bneck' else cout     # st_gcn_mstcn_1x1.py:192
            self.tcn_start = nn.Sequential(nn.BatchNorm2d(cout), nn.ReLU(inplace=False))
            if tcn == 'bneck':
                self.conv_1x1_start = nn.Conv2d(cout, width, (1, 1))
            self.tcn_1 = nn.Conv2d(width, width, (3, 1), (stride, 1), (1, 0))
            self.tcn_2 = nn.Conv2d(width, width, (9, 1), (stride, 1), (4, 0))
            self.tcn_3 = nn.Conv2d(width, width, (15, 1), (stride, 1), (7, 0))
            if tcn == 'bneck':
                self.conv_1x1_end = nn.Conv2d(width, cout, (1, 1))
            self.tcn_end = nn.Sequential(nn.BatchNorm2d(cout), nn.Dropout(dropout))
        if dead_linear:
            self.linear = nn.Linear(3, cout)                          # st_gcnold.py:178, never used
        self.res_mode = 'none' if not residual else ('id' if (cin == cout and stride == 1) else 'conv')
        if self.res_mode == 'conv':
            self.residual = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=(stride, 1)), nn.BatchNorm2d(cout))

    def forward(self, x, adj, mst=None):
        res = 0 if self.res_mode == 'none' else (x if self.res_mode == 'id' else self.residual(x))
        g = self.gcn(x, adj)
        if self.tcn_kind == 'single':
            y = self.tcn(g)
        else:
            u = self.tcn_start(g)
            if self.tcn_kind == 'bneck':
                u = self.conv_1x1_start(u)
            y = self.tcn_1(u) * mst[0] + self.tcn_2(u) * mst[1] + self.tcn_3(u) * mst[2]
            if self.tcn_kind == 'multi3':
                y = y / 3                                             # st_gcn_mstcn.py:245
            if self.tcn_kind == 'bneck':
                y = self.conv_1x1_end(y)
            y = self.tcn_end(y)
        return F.relu(y + res)


class RefModel(nn.Module):
    """Model(in_channels, num_class, graph_args, edge_importance_weighting, **kwargs) of
    st_gcnold.py:31-69 and siblings; `kind` selects the variant."""

    def __init__(self, kind, in_channels, num_class, graph_args, edge_importance_weighting, **kwargs):
        super().__init__()
        gcn, tcn, plan, _ = KINDS[kind]
        self.kind, self.gcn_kind, self.tcn_kind = kind, gcn, tcn
        self.graph = GraphRef(**graph_args)
        if gcn == 'incep':                                            # st_gcn_msgcn.py:36-39
            self.register_buffer('A2', torch.tensor(self.graph.A2, dtype=torch.float32))
            self.register_buffer('A3', torch.tensor(self.graph.A3, dtype=torch.float32))
        self.register_buffer('A', torch.tensor(self.graph.A, dtype=torch.float32))
        K, V = self.A.size(0), self.A.size(1)
        self.data_bn = nn.BatchNorm1d(in_channels * V)
        kw0 = {k: v for k, v in kwargs.items() if k != 'dropout'}
        blocks = []
        for idx, (cin, cout, stride) in enumerate(plan):
            if idx == 0:
                blocks.append(RefBlock(kind, in_channels, cout, K, 1, residual=False, **kw0))
            else:
                blocks.append(RefBlock(kind, cin, cout, K, stride, **kwargs))
        self.st_gcn_networks = nn.ModuleList(blocks)
        nb = len(blocks)
        self.n_imp = 3 if gcn in ('incep', '3a') else 1
        names = ['edge_importance', 'edge_importance2', 'edge_importance3'][:self.n_imp]
        shapes = [self.A.size(), (self.A2.size() if gcn == 'incep' else self.A.size()),
                  (self.A3.size() if gcn == 'incep' else self.A.size())]
        for nm, shp in zip(names, shapes):
            if edge_importance_weighting:
                setattr(self, nm, nn.ParameterList([nn.Parameter(torch.ones(shp)) for _ in range(nb)]))
            else:
                setattr(self, nm, [1] * nb)
        if tcn != 'single':
            self.mstcn_importance = nn.ParameterList([nn.Parameter(torch.ones(3)) for _ in range(nb)])
        self.fcn = nn.Conv2d(256, num_class, kernel_size=1)

    def _normalise_input(self, x):                                    # st_gcnold.py:74-80
        N, C, T, V, M = x.size()
        x = x.permute(0, 4, 3, 1, 2).contiguous().view(N * M, V * C, T)
        x = self.data_bn(x)
        return x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)

    def _adj(self, i):
        if self.gcn_kind == 'plain':
            return (self.A * self.edge_importance[i],)
        if self.gcn_kind == 'incep':
            return (self.A * self.edge_importance[i], self.A2 * self.edge_importance2[i],
                    self.A3 * self.edge_importance3[i])
        return (self.A, self.edge_importance[i], self.edge_importance2[i], self.edge_importance3[i])

    def _trunk(self, x):
        x = self._normalise_input(x)
        for i, blk in enumerate(self.st_gcn_networks):
            mst = self.mstcn_importance[i] if self.tcn_kind != 'single' else None
            x = blk(x, self._adj(i), mst)
        return x

    def forward(self, x):                                             # st_gcnold.py:71-96
        N, M = x.size(0), x.size(4)
        x = self._trunk(x)
        x = F.avg_pool2d(x, x.size()[2:])
        x = x.view(N, M, -1, 1, 1).mean(dim=1)
        return self.fcn(x).view(N, -1)

    def extract_feature(self, x):                                     # st_gcnold.py:98-120
        N, M = x.size(0), x.size(4)
        x = self._trunk(x)
        _, c, t, v = x.size()
        feature = x.view(N, M, c, t, v).permute(0, 2, 3, 4, 1)
        out = self.fcn(x).view(N, M, -1, t, v).permute(0, 2, 3, 4, 1)
        return out, feature


def train_step(model, optimizer, data, label):
    """One iteration of REC_Processor.train, processor/recognition.py:249-296."""
    model.train()
    output = model(data.float())
    loss = F.cross_entropy(output, label.long())
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), output.detach()


def make_optimizer(model, base_lr=0.1, weight_decay=1e-4, nesterov=True):
    """processor/recognition.py:152-159 (SGD branch)."""
    return torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, nesterov=nesterov,
                           weight_decay=weight_decay)


def weights_init_(model, seed=None):
    """processor/recognition.py:31-44 restated: Conv2d ~ N(0,.02), bias 0; BatchNorm weight ~ N(1,.02), bias 0."""
    if seed is not None:
        torch.manual_seed(seed)
    for m in model.modules():
        if type(m) is nn.Conv2d or m.__class__.__name__.find('Conv1d') != -1:
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.fill_(0)
        elif m.__class__.__name__.find('BatchNorm') != -1:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)
    return model
