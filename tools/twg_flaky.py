"""Which side differs when tools/twg_bench.py reports a large A/B difference: repeats of the weight gradient without and with
dbias (ISTGCN_TWG_LEAN=0 in the environment: the round-1/2/3 kernels) against each other and against torch's conv2d weight gradient (fp32 on the GPU).
usage: twg_flaky.py [channels] [T] [taps] [stride] [repeats] [bf16|f16|f32]"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import istgcn_amd  # noqa
from istgcn_amd import ops
c, T, k, s, reps = (int(a) for a in (sys.argv[1:6] + ['256', '150', '15', '2', '6'][len(sys.argv) - 1:]))
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[sys.argv[6] if len(sys.argv) > 6 else 'bf16']
d, NM, V = torch.device('cuda:0'), 128, 25
torch.manual_seed(0)
Tz = (T + s - 1) // s
dz = (torch.randn(NM, Tz, V, c, device=d) * 0.1).to(dt)
g = torch.randn(NM, T, V, c, device=d).to(dt)
taps, im = ops.conv_taps_fwd(k, s)
pre = torch.stack([0.5 + torch.rand(c), 0.3 * torch.randn(c)]).to(d)
# reference: autograd of conv2d on u = relu(bn(g)) rounded to bf16, in chunks of sequences
u = torch.relu(g.float() * pre[0] + pre[1]).to(dt).float()
ref = torch.zeros(k, c, c, device=d)
for n0 in range(0, NM, 16):
    W = torch.zeros(c, c, k, 1, device=d, requires_grad=True)
    z = F.conv2d(u[n0:n0 + 16].permute(0, 3, 1, 2), W, None, stride=(s, 1), padding=((k - 1) // 2, 0))
    z.backward(dz[n0:n0 + 16].float().permute(0, 3, 1, 2))
    ref += W.grad[:, :, :, 0].permute(2, 0, 1)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
for name, wb in (('without dbias', False), ('with dbias', True)):
    outs = []
    for i in range(reps):
        outs.append(ops.tconv_wgrad(dz, g, taps, in_mul=im, pre=pre, pre_relu=True, want_bias=wb)[0].clone())
        # something else on the card between the calls
        torch.randn(1 << 22, device=d).sum().item()
    print('%-20s vs torch: %s | vs its first run: %s' % (name, ' '.join('%.1e' % rel(o, ref) for o in outs),
                                                       ' '.join('%.1e' % rel(o, outs[0]) for o in outs[1:])), flush=True)
    bad = [i for i, o in enumerate(outs) if rel(o, ref) > 1e-4]
    for i in bad[:2]:
        e = (outs[i] - ref).abs()
        j = int(e.flatten().argmax())
        tap, o_, i_ = j // (c * c), (j // c) % c, j % c
        per_tap = [float(e[t].max() / ref.abs().max()) for t in range(k)]
        print('   run %d: worst entry tap %d o %d i %d; per-tap max error: %s' % (i, tap, o_, i_, ' '.join('%.0e' % x for x in per_tap)))
