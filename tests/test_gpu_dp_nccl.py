"""GPU, world_size = 2, backend 'nccl' (= RCCL over xGMI): the overlapped two-bucket gradient exchange of harness.FlatSGD with
the HIP update kernel, one process per GPU -- the N > 1 path of bench.py (replaces nn.DataParallel, processor/my_io.py:86-87).
SKIPPED on a box with one GPU (the gpurun boxes); runs wherever the suite sees two devices.  The CPU twin of this test
(same toy model, gloo, torch restatement of the update) is tests/test_dp_gloo.py::test_overlapped_two_bucket_exchange_equals_one_bucket."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Deep(nn.Module):
    def __init__(self):
        super().__init__()
        self.l0 = nn.Linear(6, 7)
        self.dead = nn.Linear(3, 2)          # never used: its gradient stays None (the reference's dead `linear.*`)
        self.l1 = nn.Linear(7, 9)
        self.l2 = nn.Linear(9, 8)
        self.l3 = nn.Linear(8, 4)

    def forward(self, x):
        return self.l3(torch.relu(self.l2(torch.relu(self.l1(torch.relu(self.l0(x)))))))


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    import istgcn_amd  # noqa: F401
    from istgcn_amd.dp import FlatGradSync, shard_batch
    from istgcn_amd.harness import FlatSGD, train_step
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randint(0, 4, (8,), generator=g)
    xs, ys = shard_batch(X, Y, rank, world)
    xs, ys = xs.to(dev), ys.to(dev)
    res = {}
    for tag, overlap in (('two', True), ('one', False)):
        torch.manual_seed(100 + rank)             # different init per rank: the broadcast must fix it
        m = Deep().to(dev)
        sync = FlatGradSync(m)
        opt = FlatSGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, overlap=overlap,
                      early_fraction=0.5).attach_sync(sync)          # update_fn None: the HIP kernel istgcn_sgd_step
        for _ in range(4):
            train_step(m, opt, xs, ys, sync)
        torch.cuda.synchronize()
        res[tag] = {k: v.detach().cpu().clone() for k, v in m.named_parameters()}
        res[tag + '_launches'] = opt.early_launches
        res[tag + '_early'] = (opt._early_n, opt._early_end, opt.G.numel())
    res['backend'] = dist.get_backend()
    res['world'] = dist.get_world_size()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_two_bucket_exchange_on_two_gpus_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (one process per GPU over RCCL); this box has %d' % torch.cuda.device_count())
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0['backend'] == 'nccl' and r0['world'] == 2
    assert r0['two_launches'] == 3 and r1['two_launches'] == 3 and r0['one_launches'] == 0    # early bucket from step 2 on
    n_early, end_early, total = r0['two_early']
    assert 0 < n_early and 0 < end_early < total
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
        assert torch.equal(r0['two'][k], r1['two'][k]), k                         # the ranks stay in lock step
        assert torch.allclose(r0['two'][k], r0['one'][k], atol=1e-6), k           # two buckets == one bucket
        assert torch.allclose(r0['two'][k], p.detach(), atol=5e-6), k             # == full-batch torch.optim.SGD
