"""Which Python call sites launch the small framework kernels (memcpy / fill / copy) of one training step?"""
import importlib, os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import istgcn_amd  # noqa
from istgcn_amd import harness
from bench import MODELS
gargs, nc, V = MODELS['st_gcn_msgcn']
m = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(3, nc, gargs, True, dropout=0.5, compute_dtype=torch.bfloat16)
m.apply(harness.weights_init); m.cuda().train()
opt = harness.make_optimizer(m)
x = torch.randn(8, 3, 300, V, 2).cuda(); y = torch.randint(0, nc, (8,)).cuda()
for _ in range(3):
    harness.train_step(m, opt, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.train_step(m, opt, x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=45, max_name_column_width=60))
