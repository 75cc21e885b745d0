"""Data parallelism for the drop-in Models: one process per GPU, batch sharded across ranks, ONE flat fp32 gradient
bucket all-reduced per step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference's only multi-GPU mechanism is single-process `nn.DataParallel` (processor/my_io.py:86-87): replicas see
N/p clips each, BatchNorm statistics stay per replica, gradients are summed onto GPU 0.  Here every rank keeps its own
BatchNorm statistics likewise, and the mean-over-ranks of per-rank mean-loss gradients equals the full-batch mean
gradient of recognition.py:278 for equal shards (drop_last, processor/processor.py:74).

Why one bucket: the models have 122-246 parameter tensors totalling 3.6-31 MB; on 7x153 GB/s xGMI
links a ring all-reduce of 31 MB is ~0.4 ms, so per-tensor collectives would be pure launch latency.  Parameters
whose gradient is None (the dead `linear.*` / `gcn.branch.bn.*` of the reference) are left out, identically on all
ranks because it is a property of the model code, not of the data.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, model, group=None, broadcast=True):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._live = None
        self._flat = None
        if broadcast and self.world > 1:
            self.broadcast_state()

    def broadcast_state(self, src=0):
        """rank `src`'s parameters and buffers to every rank (DataParallel replicates from device 0 every forward)."""
        with torch.no_grad():
            tensors = [t for t in list(self.model.parameters()) + list(self.model.buffers())
                       if torch.is_floating_point(t)]
            if not tensors:
                return
            flat = torch.cat([t.reshape(-1).float() for t in tensors])
            dist.broadcast(flat, src, group=self.group)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
                off += n

    def __call__(self):
        """Average gradients over ranks (call between backward() and optimizer.step())."""
        if self.world == 1:
            return
        if self._live is None:
            self._live = [i for i, p in enumerate(self.params) if p.grad is not None]
            n = sum(self.params[i].numel() for i in self._live)
            dev = self.params[self._live[0]].grad.device
            self._flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        views = []
        for i in self._live:
            g = self.params[i].grad
            if g is None:
                raise RuntimeError('FlatGradSync: parameter %d had a gradient on the first step and has none now' % i)
            n = g.numel()
            views.append((g, off, n))
            off += n
        torch._foreach_copy_([self._flat[o:o + n].view_as(g) for g, o, n in views], [g for g, _, _ in views])
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(self.world)
        torch._foreach_copy_([g for g, _, _ in views], [self._flat[o:o + n].view_as(g) for g, o, n in views])

    def all_reduce_flat_(self, flat, async_op=False):
        """SUM-all-reduce a caller-owned flat gradient buffer (a contiguous slice of harness.FlatSGD's `G`) in place: the
        whole exchange step of the data-parallel path, no packing.  The caller folds 1/world into its update.
        async_op=True returns the collective's Work handle (FlatSGD launches the early bucket from the backward pass and
        waits for it in step(): with RCCL the collective runs on its own stream next to the rest of the backward kernels)."""
        if self.world > 1:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                return work
        return None if async_op else flat

    def agree_on(self, ints, device):
        """rank 0's list of ints on every rank (FlatSGD's layout order: all ranks must cut the flat buffer identically)"""
        if self.world <= 1:
            return list(ints)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        n = torch.tensor([len(ints)], dtype=torch.int64, device=device)
        dist.broadcast(n, src, group=self.group)
        t = torch.tensor(list(ints) if int(n) == len(ints) else [0] * int(n), dtype=torch.int64, device=device)
        dist.broadcast(t, src, group=self.group)
        return [int(v) for v in t.cpu()]

    @property
    def bucket_bytes(self):
        return 0 if self._flat is None else self._flat.numel() * 4


def shard_batch(data, label, rank, world):
    """Equal contiguous shards of a global batch (DataParallel's scatter along dim 0)."""
    n = data.shape[0]
    if n % world:
        raise ValueError('global batch %d not divisible by world size %d (the reference uses drop_last)' % (n, world))
    per = n // world
    return data[rank * per:(rank + 1) * per], label[rank * per:(rank + 1) * per]
