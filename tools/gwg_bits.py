"""Graph-conv weight gradient (ops.gcn_wgrad) on a list of shapes -> one sha256 per (shape, output): run it under two builds of
the library (ISTGCN_LIB_PATH) and diff the output to show that a kernel change left the results bit-identical.
Shapes cover: 64 / 128 / 256 channels, K = 1..4, V = 18 / 25 / 32, frame counts that are no multiple of the 4-frame batch,
fewer batches than workgroups, both 16-bit types."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import istgcn_amd  # noqa: F401
from istgcn_amd import ops

d = torch.device('cuda:0')
CASES = [  # NM, T, V, Cin, Cout, K
    (128, 150, 25, 128, 128, 3), (128, 75, 25, 256, 256, 3), (128, 300, 25, 64, 64, 3), (16, 150, 25, 64, 128, 3),
    (3, 7, 25, 64, 64, 3), (1, 1, 25, 64, 64, 3), (5, 13, 18, 128, 64, 3), (2, 9, 32, 64, 192, 2), (7, 11, 25, 64, 128, 4),
    (3, 50, 25, 128, 256, 1), (256, 75, 18, 128, 128, 3), (2, 2, 3, 64, 64, 1)]
for dt in (torch.bfloat16, torch.float16):
    for (NM, T, V, Cin, Cout, K) in CASES:
        g = torch.Generator(device='cpu').manual_seed(NM * 1000 + T * 10 + K)
        x = torch.randn(NM, T, V, Cin, generator=g).to(d).to(dt)
        dy = torch.randn(NM, T, V, Cout, generator=g).to(d).to(dt)
        A = torch.rand(K, V, V, generator=g).to(d)
        outs = []
        for rep in range(2):
            dW, S = ops.gcn_wgrad(dy, x, A)
            torch.cuda.synchronize()
            outs.append((dW.cpu().numpy().tobytes(), S.cpu().numpy().tobytes()))
        ref = (torch.einsum('ntwc,kvw,ntvi->kci', dy.float(), A.to(dt).float(), x.float()))
        err = float((dW - ref).abs().max() / ref.abs().max())
        Sref = dy.float().sum((0, 1))
        errS = float((S - Sref).abs().max() / Sref.abs().max())
        print('%s NM=%d T=%d V=%d %d->%d K=%d  dW %s  S %s  rerun-identical %s  rel-err-vs-torch %.1e  S-err %.1e' % (
            str(dt)[6:], NM, T, V, Cin, Cout, K, hashlib.sha256(outs[0][0]).hexdigest()[:16], hashlib.sha256(outs[0][1]).hexdigest()[:16],
            outs[0] == outs[1], err, errS), flush=True)
