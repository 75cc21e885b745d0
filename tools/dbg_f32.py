import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
d = torch.device('cuda:0')
V, C, K = 25, 64, 1
x = torch.zeros(1, 2, V, C, device=d)
for v in range(V):
    for c in range(C):
        x[0, 0, v, c] = 100 * v + c
        x[0, 1, v, c] = -(100 * v + c)
W = torch.eye(C, device=d).view(C, 1, C)
A = torch.eye(V, device=d).view(1, V, V)
y = ops.gcn_forward(x, A, ops.pack_gcn_weight(W, torch.float32), C)
torch.cuda.synchronize()
print('identity: max err', float((y - x).abs().max()))
print('y[0,0,:6,:6]\n', y[0, 0, :6, :6].cpu())
print('y[0,0,:6,30:36]\n', y[0, 0, :6, 30:36].cpu())
# permuted adjacency: y[w] = x[(w+1)%V]
P = torch.zeros(V, V, device=d)
for v in range(V):
    P[(v + 1) % V, v] = 1.0
y2 = ops.gcn_forward(x, P.view(1, V, V), ops.pack_gcn_weight(W, torch.float32), C)
ref = torch.roll(x, -1, 2)
print('shift: max err', float((y2 - ref).abs().max()))
print('y2[0,0,:6,:4]\n', y2[0, 0, :6, :4].cpu())
