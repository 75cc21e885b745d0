"""CPU, world_size=2, gloo: the flat-bucket gradient all-reduce wiring (the N>1 path of bench.py).
The kernels are not involved: a small torch model stands in for the Model; what is tested is that the averaged
gradients equal the full-batch gradient, dead parameters are skipped, and state broadcast works."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.dead = nn.Linear(3, 2)          # never used: grad stays None, like st_gcnold's `linear`
        self.b = nn.Linear(5, 4)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from istgcn_amd.dp import FlatGradSync, shard_batch
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)             # different init per rank: broadcast must fix it
    m = Tiny()
    sync = FlatGradSync(m)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    xs, ys = shard_batch(X, Y, rank, world)
    for _ in range(2):                        # twice: second call reuses the bucket
        m.zero_grad()
        nn.functional.cross_entropy(m(xs), ys).backward()
        sync()
    res = {k: v.grad.clone() if v.grad is not None else None for k, v in m.named_parameters()}
    res['__w'] = m.a.weight.detach().clone()
    res['__bytes'] = sync.bucket_bytes
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_full_batch():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0['__w'], r1['__w'])                       # broadcast from rank 0
    torch.manual_seed(100)
    ref = Tiny()
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    nn.functional.cross_entropy(ref(X), Y).backward()
    for k, p in ref.named_parameters():
        if p.grad is None:
            assert r0[k] is None and r1[k] is None
        else:
            assert torch.allclose(r0[k], p.grad, atol=1e-6) and torch.allclose(r1[k], p.grad, atol=1e-6)
    assert r0['__bytes'] == 4 * sum(p.numel() for k, p in ref.named_parameters() if p.grad is not None)


def test_shard_batch_rejects_ragged():
    from istgcn_amd.dp import shard_batch
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(7, 2), torch.zeros(7), 0, 2)
