"""Eval-mode inference (folded BatchNorms, cached plans: SURVEY 8 f4) against the training-mode forward of the same
model and batch.  python tools/infer_bench.py [model] [batch] [dtype]"""
import importlib, sys, time, json
import torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import istgcn_amd  # noqa
from istgcn_amd import harness
from bench import MODELS
tag = sys.argv[1] if len(sys.argv) > 1 else 'st_gcn_msgcn'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[3] if len(sys.argv) > 3 else 'bf16']
gargs, nc, V = MODELS[tag]
T = 600 if tag.endswith('deep') else 300
torch.manual_seed(0)
m = importlib.import_module('istgcn_amd.net.' + tag).Model(3, nc, gargs, True, dropout=0.5, compute_dtype=dt)
m.apply(harness.weights_init)
m.cuda()
x = torch.randn(B, 3, T, V, 2).cuda()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


m.train()
t_train = timeit(lambda: m(x))
m.eval()
with torch.no_grad():
    t_inf = timeit(lambda: m(x))
    t_feat = timeit(lambda: m.extract_feature(x)) if tag in ('st_gcnold',) else None
print(json.dumps({'model': tag, 'batch': B, 'dtype': str(dt)[6:], 'train_forward_ms': round(t_train, 3),
                  'inference_ms': round(t_inf, 3), 'speedup': round(t_train / t_inf, 2),
                  'inference_clips_per_s': round(B / t_inf * 1e3, 1), 'extract_feature_ms': t_feat}))
