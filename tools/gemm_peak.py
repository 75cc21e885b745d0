"""What the vendor GEMM (hipBLASLt through torch.matmul) sustains on this box in bf16 -- the practical ceiling next to the
2.5 PFLOP/s dense peak bench.py divides by -- on all-zero operands (no switching power) and on random ones, for a large
square problem and for the implicit-GEMM shapes of the temporal conv (M = positions, N = C_out, K = taps * C_in)."""
import torch
d = torch.device('cuda:0')


def run(M, N, K, fill):
    a = torch.zeros(M, K, device=d, dtype=torch.bfloat16)
    b = torch.zeros(K, N, device=d, dtype=torch.bfloat16)
    if fill:
        a.normal_()
        b.normal_()
    for _ in range(3):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    it = max(5, int(2e13 / (2.0 * M * N * K)))
    it = min(it, 200)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return 2.0 * M * N * K / ms / 1e9, ms


for (M, N, K, what) in ((8192, 8192, 8192, 'square 8192^3'), (16384, 16384, 4096, '16384 x 16384 x 4096'),
                        (960000, 64, 576, 'tconv 64 ch (P=960k, 9 taps)'), (480000, 128, 1152, 'tconv 128 ch (P=480k)'),
                        (240000, 256, 2304, 'tconv 256 ch (P=240k)')):
    for fill in (0, 1):
        tf, ms = run(M, N, K, fill)
        print('%-34s %-7s %8.1f TFLOP/s  %.3f ms  (%.2f of 2500)' % (what, 'random' if fill else 'zeros', tf, ms, tf / 2500), flush=True)
