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


def _torch_sgd_update(P, G, M, lr, momentum, wd, nesterov, gscale):
    """torch restatement of istgcn_sgd_step (csrc/optim.hip) for the CPU test of FlatSGD's bucket logic."""
    g = G * gscale + wd * P
    M.mul_(momentum).add_(g)
    P.add_(g + momentum * M if nesterov else M, alpha=-lr)


def _worker_flat_sgd(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from istgcn_amd.dp import FlatGradSync, shard_batch
    from istgcn_amd.harness import FlatSGD, train_step
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    m = Tiny()
    sync = FlatGradSync(m)
    opt = FlatSGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4,
                  update_fn=_torch_sgd_update).attach_sync(sync)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    xs, ys = shard_batch(X, Y, rank, world)

    class Wrap(nn.Module):                     # harness.train_step feeds (N, ...) float data straight to the model
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            return self.inner(x)
    w = Wrap(m)
    for _ in range(3):
        train_step(w, opt, xs, ys, sync)      # sync is the optimizer's own: all-reduced in place on its flat buffer
    res = {k: v.detach().clone() for k, v in m.named_parameters()}
    res['__bytes'] = opt.bucket_bytes
    res['__is_view'] = all(p.data_ptr() >= opt.P.data_ptr() and p.data_ptr() < opt.P.data_ptr() + opt.P.numel() * 4
                           for k, p in m.named_parameters() if not k.startswith('dead'))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_flat_sgd_allreduces_its_own_bucket_and_matches_full_batch_sgd():
    """f1 (recognition.py:154-159,287-289): parameters / gradients / momentum as views of three flat buffers, the
    all-reduce in place on the gradient buffer, 1/world folded into the update; two ranks with half the batch each must
    follow the single-process full-batch torch.optim.SGD trajectory; dead parameters stay untouched."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_flat_sgd, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    torch.manual_seed(100)
    ref = Tiny()
    init_dead = {k: v.detach().clone() for k, v in ref.named_parameters() if k.startswith('dead')}
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    for _ in range(3):
        opt.zero_grad()
        nn.functional.cross_entropy(ref(X), Y).backward()
        opt.step()
    for k, p in ref.named_parameters():
        assert torch.allclose(r0[k], p.detach(), atol=2e-6), k
        assert torch.equal(r0[k], r1[k]), k
    for k, v in init_dead.items():
        assert torch.equal(r0[k], v)                                   # no gradient -> no update, no weight decay
    assert r0['__is_view'] and r0['__bytes'] == 4 * sum((p.numel() + 3) // 4 * 4 for k, p in ref.named_parameters()
                                                        if not k.startswith('dead'))
