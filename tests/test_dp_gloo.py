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


# ---------------------------------------------------------------------------------------------------------------
# overlapped exchange: the early bucket (first-final gradients) is all-reduced from inside the backward pass
# ---------------------------------------------------------------------------------------------------------------
class Deep(nn.Module):
    def __init__(self):
        super().__init__()
        self.l0 = nn.Linear(6, 7)
        self.dead = nn.Linear(3, 2)
        self.l1 = nn.Linear(7, 9)
        self.l2 = nn.Linear(9, 8)
        self.l3 = nn.Linear(8, 4)

    def forward(self, x):
        return self.l3(torch.relu(self.l2(torch.relu(self.l1(torch.relu(self.l0(x)))))))


def _worker_overlap(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from istgcn_amd.dp import FlatGradSync, shard_batch
    from istgcn_amd.harness import FlatSGD, train_step
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    xs, ys = shard_batch(X, Y, rank, world)
    res = {}
    for tag, overlap in (('two', True), ('one', False)):
        torch.manual_seed(100 + rank)
        m = Deep()
        sync = FlatGradSync(m)
        opt = FlatSGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4,
                      update_fn=_torch_sgd_update, overlap=overlap, early_fraction=0.5).attach_sync(sync)
        for _ in range(4):
            train_step(m, opt, xs, ys, sync)
        res[tag] = {k: v.detach().clone() for k, v in m.named_parameters()}
        res[tag + '_launches'] = opt.early_launches
        res[tag + '_early'] = (opt._early_n, opt._early_end, opt.G.numel())
        res[tag + '_order'] = list(opt._order)
        if overlap:
            # a second backward before step() would add local gradients onto already reduced ones: refused
            opt.zero_grad()
            nn.functional.cross_entropy(m(xs), ys).backward()
            try:
                nn.functional.cross_entropy(m(xs), ys).backward()
                res['second_backward'] = 'accepted'
            except RuntimeError as e:
                res['second_backward'] = str(e)
            opt.step()                                       # (both ranks: keeps the collectives paired)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_two_bucket_exchange_equals_one_bucket():
    """flat buffer laid out in gradient-arrival order, its early prefix all-reduced asynchronously from the
    backward pass (hook of the prefix's last parameter), the suffix in step(): same trajectory as the single blocking
    all-reduce and as full-batch torch SGD; the early collective really is launched from step 2 on."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0['two_launches'] == 3 and r1['two_launches'] == 3          # steps 2..4 (step 1 fixes the layout)
    assert r0['one_launches'] == 0
    n_early, end_early, total = r0['two_early']
    assert 0 < n_early and 0 < end_early < total and end_early >= 0.5 * total
    # arrival order: the last layer's parameters first (backward reaches them first), the first layer's last
    names = [k for k, _ in Deep().named_parameters()]
    req = [k for k in names]
    first, last = req[r0['two_order'][0]], req[r0['two_order'][-1]]
    assert first.startswith('l3.') and last.startswith('l0.'), (first, last)
    assert r0['two_order'] == r1['two_order']
    assert 'second backward' in r0['second_backward']
    torch.manual_seed(100)
    ref = Deep()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    for _ in range(4):
        opt.zero_grad()
        nn.functional.cross_entropy(ref(X), Y).backward()
        opt.step()
    for k, p in ref.named_parameters():
        assert torch.equal(r0['two'][k], r1['two'][k]), k
        assert torch.allclose(r0['two'][k], r0['one'][k], atol=1e-7), k
        assert torch.allclose(r0['two'][k], p.detach(), atol=3e-6), k


def test_flat_sgd_momentum_resume_across_layout_orders():
    """The flat momentum is saved with its layout order; an optimizer whose layout came out in another order (hooks
    unavailable, another torch version) installs it parameter by parameter."""
    from istgcn_amd.harness import FlatSGD
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)

    def run(m, opt, n):
        for _ in range(n):
            opt.zero_grad()
            nn.functional.cross_entropy(m(X), Y).backward()
            opt.step()
    torch.manual_seed(5)
    ma = Deep()
    oa = FlatSGD(ma.parameters(), lr=0.1, update_fn=_torch_sgd_update)
    run(ma, oa, 2)
    sd_model = {k: v.clone() for k, v in ma.state_dict().items()}
    sd_opt = oa.state_dict()
    assert sd_opt['order'][0] != 0                              # arrival order, not model order
    run(ma, oa, 2)
    mb = Deep()
    mb.load_state_dict(sd_model)
    ob = FlatSGD(mb.parameters(), lr=0.1, update_fn=_torch_sgd_update)
    for h in ob._hooks:                                         # no hooks: the layout falls back to model order
        h.remove()
    ob._hooks = []
    ob.load_state_dict(sd_opt)
    run(mb, ob, 2)
    assert ob._order != oa._order and ob._early_n == 0
    for (k, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.allclose(a, b, atol=1e-7), k
